"""Host-only costing of an LDS-resident copy of the hottest item rows for the user-row phase of the SpMM (VERDICT r05, item 5;
north_star's "LDS-staged embedding tiles"; the line replaced is model.py:102).

In the user-row phase every wave gathers ITEM rows; the item popularity of the SURVEY 8(d) generator is Zipf-like, so a small
set of hot items takes a large share of the gathers.  An LDS copy of the top-H item rows (H * d * 4 bytes per workgroup's CU:
H = 80 / 320 / 640 -> 20 / 80 / 160 KB at d = 64) would serve exactly the gathers whose column is among those H.  This script
counts, on the seeded Amazon-like, Yelp-like and Gowalla-like graphs:
  * the share of user-phase gathers (= train pairs) that fall on the top-H items, H = 80 / 320 / 640 (+ 2 560 / 16 384:
    what a 4 MiB L2 slice holds of 256-byte rows, for orientation);
  * what filling the copy costs: every workgroup that wants the copy must first READ it (H rows from L2) — per CU and per
    launch H * 256 B; with 256 CUs that is 256 * H rows = an extra fraction of the phase's gathers;
  * the occupancy price: 160 KB of LDS per CU shared by the resident workgroups; a copy of S KB leaves (160 - S) KB.
No GPU is used.  Output: profiles/r06b_lds_hot_rows_costing.json"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from igcn_cf_amd.dataset import SyntheticDataset                               # noqa: E402

out = {'d': 64, 'row_bytes': 256, 'graphs': {}}
for preset in ('amazon', 'yelp', 'gowalla'):
    ds = SyntheticDataset({'name': 'SyntheticDataset', 'preset': preset, 'seed': 2021, 'device': 'cpu'})
    ta = np.asarray(ds.train_array)
    deg = np.bincount(ta[:, 1], minlength=ds.n_items).astype(np.int64)
    total = int(deg.sum())
    order = np.sort(deg)[::-1]
    cum = np.cumsum(order)
    g = {'users': ds.n_users, 'items': ds.n_items, 'user_phase_gathers': total, 'max_item_degree': int(order[0]), 'top_H': {}}
    for H in (80, 320, 640, 2560, 16384):
        share = float(cum[min(H, len(cum)) - 1] / total)
        fill_rows = 256 * H                                                     # every CU reads the H rows once per launch
        g['top_H'][str(H)] = {'lds_KB_at_d64': H * 256 / 1024, 'share_of_user_phase_gathers': share,
                              'gathers_served': int(cum[min(H, len(cum)) - 1]),
                              'fill_rows_per_launch_256_CUs': fill_rows, 'fill_over_served': fill_rows / float(cum[min(H, len(cum)) - 1]),
                              'net_gathers_saved_share': (float(cum[min(H, len(cum)) - 1]) - fill_rows) / total}
    out['graphs'][preset] = g
    print(preset, json.dumps(g['top_H'], indent=None))
path = os.path.join(ROOT, 'profiles', 'r06b_lds_hot_rows_costing.json')
json.dump(out, open(path, 'w'), indent=1)
print('wrote', path)
