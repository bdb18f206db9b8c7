"""Probe: do gloo collectives accept device tensors when two ranks share one GPU? (rehearsal of the sharded path on a 1-GPU box)"""
import os, sys, torch, torch.distributed as dist
import torch.multiprocessing as mp


def worker(rank, world, port):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    dev = torch.device('cuda', 0)
    x = torch.full((4, 3), float(rank + 1), device=dev)
    out = torch.empty(4 * world, 3, device=dev)
    res = {}
    for name, fn in (('all_gather_into_tensor', lambda: dist.all_gather_into_tensor(out, x)),
                     ('all_reduce', lambda: dist.all_reduce(x)),
                     ('reduce_scatter_tensor', lambda: dist.reduce_scatter_tensor(x, out.clone())),
                     ('all_gather', lambda: dist.all_gather([torch.empty_like(x) for _ in range(world)], x)),
                     ('broadcast', lambda: dist.broadcast(x, 0)),
                     ('async all_gather', lambda: dist.all_gather_into_tensor(out, x, async_op=True).wait())):
        try:
            fn(); torch.cuda.synchronize(); res[name] = 'ok'
        except Exception as e:
            res[name] = repr(e)[:120]
    if rank == 0:
        print(res, out[:, 0].tolist())
    dist.destroy_process_group()


if __name__ == '__main__':
    mp.spawn(worker, args=(2, 29533), nprocs=2)
