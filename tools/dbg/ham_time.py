import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
import photogrammetry_amd as pg
DEV = "cuda:0"
e = pg.Engine(0)
rng = np.random.default_rng(1)
F, N = 24, 4096
d = torch.from_numpy(rng.integers(0, 2**31, (F, N, 8), dtype=np.int64).astype(np.int32)).to(DEV)
i32 = dict(dtype=torch.int32, device=DEV)
cnt = torch.full((F,), N, **i32)
pl = [(a, b) for a in range(F) for b in range(a + 1, F)][:256]
d_pl = torch.tensor(pl, **i32); out = torch.zeros((len(pl), N, 3), **i32)
e.profile_serialize(True)
for rep in range(3):
    e.profile_reset(); e.profile_enable(True)
    e.match_batch_dev(d, cnt, N, 8, d_pl, len(pl), out, max_count=N)
    torch.cuda.synchronize(); e.profile_enable(False)
    try: e.check_status()
    except Exception as ex: print("status:", ex)
    print({k: round(e.profile_get(k)[1], 3) for k in ("ham_argmin",)}, e.profile_get("ham_argmin")[0])
