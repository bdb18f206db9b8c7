set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout -k 10 600 python -m pytest tests/test_models_gpu.py tests/test_model_golden.py -x -q > $O/r03u_tests.log 2>&1 || { tail -40 $O/r03u_tests.log; exit 1; }
tail -3 $O/r03u_tests.log
python - <<'PY'
import json, sys, time, torch
sys.path.insert(0, '.')
from igcn_cf_amd import config as cfg
from igcn_cf_amd.dataset import get_dataset
from igcn_cf_amd.model import get_model
from igcn_cf_amd.trainer import get_trainer
dev = torch.device('cuda')
for preset in ('yelp', 'amazon'):
    ds_cfg, m_cfg, t_cfg = cfg.get_synthetic_config(dev, preset)[2]
    ds = get_dataset(ds_cfg)
    out = {'preset': preset}
    for tag, fused in (('separate_losses', False), ('fused_step', True), ('separate_again', False), ('fused_again', True)):
        torch.manual_seed(2021)
        model = get_model(m_cfg, ds)
        trainer = get_trainer(dict(t_cfg, fused_inmo_step=fused), ds, model)
        model.train()
        it = zip(trainer.sampler.epoch_node_batches(trainer.batch_size, model.n_users), trainer.aux_sampler.epoch_batches(trainer.batch_size))
        for _ in range(8):
            trainer.igcn_node_step(*next(it))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(80):
            trainer.igcn_node_step(*next(it))
        torch.cuda.synchronize()
        out[tag] = round((time.perf_counter() - t0) * 1e3 / 80, 4)
    print(json.dumps(out), flush=True)
PY
