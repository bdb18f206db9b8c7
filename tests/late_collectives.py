"""Test harness shared by tests/test_dist_cpu.py and tests/test_dist_gpu.py: the collectives of `world` ranks that run as THREADS
of one process and complete as late as RCCL's may (see LateCollectives), and the thread runner.  Works on host and device
tensors alike (copies, fills and comparisons go through torch)."""
import torch


class LateCollectives:
    """The collectives of `world` ranks that run as THREADS of this process, completing as late as RCCL's may.

    gloo's `work.wait()` blocks the host until the copy is done; RCCL's only orders streams — the copy lands at some point
    between the call and the completion of the wait.  This object takes the adversarial legal schedule:
      * at the CALL of an all-gather the destination is poisoned (NaN): whoever reads it before the wait reads garbage;
        the source is snapshotted;
      * a rank's copy happens inside ITS wait(), once every rank has issued the same collective, from the sources as they are
        THEN — a source overwritten before its collective completed is detected (compared with the snapshot) and reported;
      * a wait() returns only when every rank has taken its copy (a source is free again when its owner's wait returns).
    An all-gather whose wait the code under test forgets (`skip_wait`, the negative control) never lands on that rank: its
    destination stays poisoned, the other ranks are not kept waiting for it.  all_reduce is blocking and in place (as in the code
    under test): every rank contributes once, the sums replace the buffers when all have arrived."""

    def __init__(self, world, timeout=60.0):
        import threading
        self.world, self.timeout = world, timeout
        self.cv = threading.Condition()
        self.pending = {}                  # sequence number -> {rank: (out, inp, snapshot)}
        self.copied = {}                   # sequence number -> ranks that took their copy (or will never take it)
        self.issued = [0] * world
        self.reduce_slots, self.reduce_read = {}, {}
        self.reduced = [0] * world
        self.errors = []
        self.n_async = 0
        self.skip_wait = None              # (rank, nth asynchronous all-gather of that rank): its wait() is a no-op

    def bind(self, rank):
        return RankView(self, rank)

    def _until(self, cond):
        with self.cv:
            if not self.cv.wait_for(cond, self.timeout):
                raise TimeoutError('a rank never reached the collective the others are in')


class Work:
    def __init__(self, owner, rank, seq):
        self.owner, self.rank, self.seq = owner, rank, seq

    def wait(self):
        o = self.owner
        o._until(lambda: len(o.pending.get(self.seq, {})) == o.world)      # every rank has issued this collective
        entries = o.pending[self.seq]
        out = entries[self.rank][0]
        rows = entries[self.rank][1].shape[0]
        for q in range(o.world):
            _, inp, snap = entries[q]
            if not torch.equal(inp, snap):
                with o.cv:
                    o.errors.append('rank %d overwrote the source of all-gather %d before it completed' % (q, self.seq))
            out[q * rows:(q + 1) * rows].copy_(inp)
        with o.cv:
            o.copied.setdefault(self.seq, set()).add(self.rank)
            o.cv.notify_all()
        o._until(lambda: len(o.copied[self.seq]) == o.world)               # every rank has its copy: the sources are free again
        return True


class RankView:
    def __init__(self, owner, rank):
        self.owner, self.rank, self.n_async = owner, rank, 0

    def active(self):
        return True

    def all_gather_into_tensor(self, out, inp, async_op):
        o = self.owner
        assert out.shape[0] == inp.shape[0] * o.world and out.is_contiguous() and inp.is_contiguous()
        out.fill_(float('nan'))                                 # the collective may write its destination from now on
        forget = bool(async_op) and o.skip_wait == (self.rank, self.n_async)
        with o.cv:
            seq = o.issued[self.rank]
            o.issued[self.rank] += 1
            o.pending.setdefault(seq, {})[self.rank] = (out, inp, inp.clone())
            o.n_async += bool(async_op)
            if forget:
                o.copied.setdefault(seq, set()).add(self.rank)  # (this rank will never take its copy)
            o.cv.notify_all()
        work = Work(o, self.rank, seq)
        if not async_op:
            work.wait()
            return None
        self.n_async += 1
        return type('ForgottenWait', (), {'wait': lambda self: True})() if forget else work

    def all_reduce(self, buf):
        o = self.owner
        with o.cv:
            seq = o.reduced[self.rank]
            o.reduced[self.rank] += 1
            o.reduce_slots.setdefault(seq, {})[self.rank] = buf
            o.cv.notify_all()
        o._until(lambda: len(o.reduce_slots[seq]) == o.world)
        total = sum(o.reduce_slots[seq][q].clone() for q in range(o.world))
        with o.cv:
            o.reduce_read.setdefault(seq, set()).add(self.rank)
            o.cv.notify_all()
        o._until(lambda: len(o.reduce_read[seq]) == o.world)             # everybody has read everybody's contribution
        buf.copy_(total)


def run_ranks(world, fn):
    """fn(rank) on one thread per rank; exceptions of any rank are re-raised here."""
    import threading
    results, errors = [None] * world, []

    def body(r):
        try:
            results[r] = fn(r)
        except BaseException as e:                              # noqa: BLE001 (a broken barrier in one rank must not hide the cause in another)
            errors.append((r, e))
    threads = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(120)
    if errors:
        raise errors[0][1]
    return results
