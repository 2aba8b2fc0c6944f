"""Per-wave stamps of wbc_step_kernel launches INSIDE the bench's loop (the bench's env, runner and random-init policy; wave-timing
build): SIMD busy mean / max, how the heavy waves are spread, how good the deal's hints are.  usage: python tools/wave_bench_state.py [N]"""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-whole-body-control_amd")); sys.path.insert(0, ROOT)
os.environ["WBC_AMD_LIB"] = os.path.join(ROOT, "deep-whole-body-control_amd", "wbc_amd", "libwbc_amd_" + os.environ.get("WBC_WAVE_LIB", "wavetiming") + ".so")
import numpy as np, torch
from wbc_amd.config import WidowGo1RoughCfg, WidowGo1RoughCfgPPO, class_to_dict
from wbc_amd.envs import WidowGo1
from wbc_amd.rsl_rl.runners import OnPolicyRunner
from wbc_amd.native import lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = WidowGo1RoughCfg(); cfg.env.num_envs = n; cfg.terrain.mesh_type = "plane"
tc = WidowGo1RoughCfgPPO(); torch.manual_seed(tc.seed)
env = WidowGo1(cfg, sim_device="cuda:0", seed=tc.seed)
runner = OnPolicyRunner(env, class_to_dict(tc), log_dir=None, device="cuda:0")
env.collect_episode_stats = True
runner.learn(2, init_at_random_ep_len=True)
runner.learn(5)
L = lib(); L.wbc_debug_set_wave_timing.argtypes = [C.c_void_p]
buf = torch.zeros(6 * n, dtype=torch.int64, device="cuda")
L.wbc_debug_set_wave_timing(buf.data_ptr())
raw = env.sim.step
recs = []
def rec_step(*a, **kw):
    r = raw(*a, **kw)
    if len(recs) < 80:
        torch.cuda.synchronize(); recs.append(buf.cpu().numpy().reshape(n, 6).copy())
    return r
env.sim.step = rec_step
runner.learn(2)
L.wbc_debug_set_wave_timing(None)
prev = None; rows = []
for t in recs:
    t0, t1, fl = t[:, 0], t[:, 1], t[:, 2]
    hw = (fl >> 32) & 0xFFFF; xcc = (fl >> 24) & 15
    sid = ((hw >> 4) & 3) | (((hw >> 8) & 255) << 2) | (xcc << 10)
    rst = (fl & 1) == 1; hint = ((fl >> 1) & 1) == 1; ncon = (fl >> 8) & 255
    heavy = (ncon > 0) | rst; dur = (t1 - t0).astype(float)
    ids, inv = np.unique(sid, return_inverse=True)
    first = np.full(len(ids), np.iinfo(np.int64).max); last = np.zeros(len(ids), dtype=np.int64)
    np.minimum.at(first, inv, t0); np.maximum.at(last, inv, t1)
    busy = (last - first).astype(float); nh = np.bincount(inv, weights=heavy.astype(float))
    row = dict(busy_mean=busy.mean(), busy_p90=np.percentile(busy, 90), busy_p99=np.percentile(busy, 99), busy_p999=np.percentile(busy, 99.9), busy_max=busy.max(), dur_p99=np.percentile((t1 - t0), 99), dur_p999=np.percentile((t1 - t0), 99.9), dur_max=float((t1 - t0).max()), span=float(t1.max() - t0.min()), heavy=heavy.mean(), rst=rst.mean(), ncon=ncon[ncon > 0].mean() if (ncon > 0).any() else 0,
               dl=dur[~heavy].mean(), dh=dur[heavy].mean(), nh4=(nh >= 3).sum())
    if prev is not None:
        ph = prev
        row.update(hit=heavy[ph].mean() if ph.any() else 0, miss=heavy[~ph].mean(), hshare=ph.mean())
    rows.append(row); prev = hint & ~rst
for k in rows[1].keys(): print(f"{k:10s} mean {np.mean([r[k] for r in rows[1:]]):10.3f}  min {np.min([r[k] for r in rows[1:]]):10.3f}  max {np.max([r[k] for r in rows[1:]]):10.3f}")
nc = np.bincount(((recs[-1][:, 2] >> 8) & 255).astype(int), minlength=12)
print("active contacts of the last launch's waves (count: waves):", " ".join(f"{i}:{c}" for i, c in enumerate(nc) if c))
# what a SIMD's busy time is made of: wave classes, and the busiest SIMDs of the last launch
import collections
acc = collections.defaultdict(list)
X = []; Y = []
for t in recs[1:]:
    t0, t1, fl = t[:, 0], t[:, 1], t[:, 2]
    hw = (fl >> 32) & 0xFFFF; xcc = (fl >> 24) & 15
    sid = ((hw >> 4) & 3) | (((hw >> 8) & 255) << 2) | (xcc << 10)
    rst = (fl & 1) == 1; ncon = (fl >> 8) & 255; dur = (t1 - t0).astype(float)
    cls = np.where(ncon == 0, 0, np.where(ncon <= 4, 1, 2)) + 3 * rst.astype(int)          # 0 light, 1 1-4 contacts, 2 5+ contacts; +3 resetting
    for c in range(6):
        if (cls == c).any(): acc[c].append((float((cls == c).mean()), dur[cls == c].mean(), np.percentile(dur[cls == c], 95)))
    ids, inv = np.unique(sid, return_inverse=True)
    first = np.full(len(ids), np.iinfo(np.int64).max); last = np.zeros(len(ids), dtype=np.int64)
    np.minimum.at(first, inv, t0); np.maximum.at(last, inv, t1)
    busy = (last - first).astype(float)
    cnt = np.stack([np.bincount(inv, weights=(cls == c).astype(float), minlength=len(ids)) for c in range(6)], 1)
    X.append(cnt); Y.append(busy)
names = ["airborne", "1-4 contacts", "5+ contacts", "airborne + reset", "1-4 contacts + reset", "5+ contacts + reset"]
for c in range(6):
    if acc[c]: a = np.array(acc[c]); print(f"{names[c]:22s}: share {a[:, 0].mean():.3f} wave duration mean {a[:, 1].mean():.0f} p95 {a[:, 2].mean():.0f}")
X = np.concatenate(X); Y = np.concatenate(Y)
coef, *_ = np.linalg.lstsq(X, Y, rcond=None)
print("SIMD busy ~ sum over its waves of:", " ".join(f"{names[c]} {coef[c]:.0f};" for c in range(6)), f"residual rms {np.sqrt(np.mean((X @ coef - Y) ** 2)):.0f}")
o = np.argsort(-Y)[:8]
print("busiest SIMDs (busy: counts per class):", " | ".join(f"{Y[i]:.0f}: {X[i].astype(int).tolist()}" for i in o))
# the slowest waves: what are they?
print("slowest waves of the last 5 launches: duration (contacts, deepest level, self-collision slots active, reset) | the SIMD's other waves' durations")
for t in recs[-5:]:
    t0, t1, fl = t[:, 0], t[:, 1], t[:, 2]
    hw = (fl >> 32) & 0xFFFF; xcc = (fl >> 24) & 15
    sid = ((hw >> 4) & 3) | (((hw >> 8) & 255) << 2) | (xcc << 10)
    dur = (t1 - t0)
    for i in np.argsort(-dur)[:6]:
        mates = np.where(sid == sid[i])[0]
        print(f"   {dur[i]} ({(fl[i] >> 8) & 255}, {(fl[i] >> 16) & 255}, {(fl[i] >> 28) & 15}, {fl[i] & 1}) | " + " ".join(f"{dur[m]}({(fl[m] >> 8) & 255},{(fl[m] >> 28) & 15},{fl[m] & 1})" for m in mates if m != i))
    ns = (fl >> 28) & 15
    print(f"   waves with self-collision slots active: {(ns > 0).sum()}, their mean duration {dur[ns > 0].mean() if (ns > 0).any() else 0:.0f}; waves with >= 7 contacts {(((fl >> 8) & 255) >= 7).sum()}")
# is the workgroup -> SIMD placement the same launch after launch inside this loop?
def placement(t):
    fl = t[:, 2]; blk = ((fl >> 48) & 0xFFFF).astype(int); hw = (fl >> 32) & 0xFFFF; xcc = (fl >> 24) & 15
    sid = ((hw >> 4) & 3) | (((hw >> 8) & 255) << 2) | (xcc << 10)
    out = np.zeros(n, dtype=np.int64); out[blk] = sid
    return out
pl = [placement(t) for t in recs]
print("fraction of workgroups on the same SIMD as in the previous launch:", " ".join(f"{(pl[i] == pl[i - 1]).mean():.2f}" for i in range(1, len(pl))))
print("... same CU:", " ".join(f"{((pl[i] >> 2) == (pl[i - 1] >> 2)).mean():.2f}" for i in range(1, min(len(pl), 20))))
print("... as in the first recorded launch:", " ".join(f"{(pl[i] == pl[0]).mean():.2f}" for i in range(1, len(pl))))
# round-robin structure: do ANY 128 consecutive workgroups of an XCD (in-XCD index j = blockIdx >> 3) land on 128 distinct SIMDs?
per = n // 8
for li in (0, 1, 2, len(pl) - 1):
    p = pl[li]; worst = {}
    for w in (32, 64, 128, 256):
        lo = 10 ** 9
        for x in range(8):
            sj = p[np.arange(per) * 8 + x]
            for st in range(0, per - w + 1, 8):
                lo = min(lo, len(set(sj[st:st + w].tolist())))
        worst[w] = lo
    print(f"launch {li}: fewest distinct SIMDs in any window of consecutive in-XCD workgroups:", worst, "| first 12 of XCD 0:", [(int(v >> 7) & 7, int(v >> 2) & 15, int(v) & 3) for v in p[np.arange(12) * 8]])
print("slowest waves, phases: total = load+substeps / post / reset / observe (contacts, self slots, reset)")
for t in recs[-4:]:
    t0, t1, fl = t[:, 0], t[:, 1], t[:, 2]
    dur = t1 - t0
    for i in np.argsort(-dur)[:5]:
        print(f"   {dur[i]} = {t[i, 3] - t0[i]} / {t[i, 4] - t[i, 3]} / {t[i, 5] - t[i, 4]} / {t1[i] - t[i, 5]}  ({(fl[i] >> 8) & 255}, {(fl[i] >> 28) & 15}, {fl[i] & 1})")
    sub = (t[:, 3] - t0).astype(float); ncon = (fl >> 8) & 255
    print("   substep phase by contacts of the last substep:", " ".join(f"{k}:{sub[ncon == k].mean():.0f}/{np.percentile(sub[ncon == k], 99):.0f}" for k in range(0, 9) if (ncon == k).sum() > 5))
