"""Which SIMD does each workgroup of wbc_step_kernel land on, launch after launch (wave-timing build), and what would a launch cost
if every SIMD got the same share of the heavy waves?  usage: python tools/wave_map.py [N]"""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-whole-body-control_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["WBC_AMD_LIB"] = os.path.join(ROOT, "deep-whole-body-control_amd", "wbc_amd", "libwbc_amd_wavetiming.so")
import numpy as np, torch
import helpers
from wbc_amd import abi
from wbc_amd.config import WidowGo1RoughCfg
from wbc_amd.native import lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
m = abi.load_default_model(); cfg = WidowGo1RoughCfg(); tc = abi.fill_task_cfg(cfg, m)
g = helpers.make_gpu(dict(model=m, wmodel=abi.fill_model(m), cfg=cfg, tcfg=tc), n, helpers.random_env_params(n, 0))
g.reset_all()
acts = [torch.randn(n, 18, device="cuda") * 0.5 for _ in range(8)]
for i in range(60): g.step(acts[i % 8])
L = lib(); L.wbc_debug_set_wave_timing.argtypes = [C.c_void_p]
buf = torch.zeros(6 * n, dtype=torch.int64, device="cuda")
L.wbc_debug_set_wave_timing(buf.data_ptr())
per = n // 8
env = np.arange(n); blk = (env % per) * 8 + env // per            # env = (b % 8) * per + b / 8
maps = []; prev_heavy = None
for rep in range(6):
    g.step(acts[rep]); torch.cuda.synchronize()
    t = buf.cpu().numpy().reshape(n, 6)
    t0, t1, fl = t[:, 0], t[:, 1], t[:, 2]
    hw = (fl >> 32) & 0xFFFF; xcc = (fl >> 24) & 15
    simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
    sid = simd | (cu << 2) | (sh << 6) | (se << 7) | (xcc << 10)
    maps.append(sid.copy())
    rst = (fl & 1) == 1; ncon = (fl >> 8) & 255
    heavy = (ncon > 0) | rst
    dur = (t1 - t0).astype(float)
    ids, inv = np.unique(sid, return_inverse=True)
    first = np.full(len(ids), np.iinfo(np.int64).max); last = np.zeros(len(ids), dtype=np.int64)
    np.minimum.at(first, inv, t0); np.maximum.at(last, inv, t1)
    busy = (last - first).astype(float)
    work = np.bincount(inv, weights=dur)
    print(f"launch {rep}: {len(ids)} SIMDs; busy mean {busy.mean():.0f} max {busy.max():.0f}; wave duration light {dur[~heavy].mean():.0f} heavy {dur[heavy].mean():.0f}; heavy share {heavy.mean():.3f}"
          + ("" if prev_heavy is None else f"; heavy now given heavy before {heavy[prev_heavy].mean():.3f}, given light before {heavy[~prev_heavy].mean():.3f}"))
    if prev_heavy is not None:
        # predictor quality: contact count of the previous launch
        pass
    nh = np.bincount(inv, weights=heavy.astype(float))
    print("   SIMDs by number of heavy waves:", " ".join(f"{k}:{int((nh == k).sum())}" for k in range(5)), "| busy by count:", " ".join(f"{k}:{busy[nh == k].mean():.0f}" for k in range(5) if (nh == k).any()))
    prev_heavy = heavy
    if rep == 0 and os.environ.get("WAVE_MAP_VERBOSE"):
        o = np.argsort(blk)
        b_sid = sid[o]                     # indexed by blockIdx
        print("   blockIdx -> (xcc, se, sh, cu, simd) of the first 40 workgroups:")
        for b in range(40): print("    ", b, int(xcc[o][b]), int(se[o][b]), int(sh[o][b]), int(cu[o][b]), int(simd[o][b]))
        for x in range(8):
            bs = np.arange(x, n, 8)
            print(f"   xcc of blocks = {x} mod 8: {sorted(set(xcc[o][bs].tolist()))}; CUs seen {len(set((sid[o][bs] >> 2).tolist()))}")
        # within XCD 0: order j -> (se, sh, cu, simd)
        bs = np.arange(0, n, 8)
        print("   XCD of block 0, in-XCD order j -> se.sh.cu.simd:", " ".join(f"{int(se[o][b])}.{int(sh[o][b])}.{int(cu[o][b])}.{int(simd[o][b])}" for b in bs[:160]))
same = [(maps[i] == maps[0]).mean() for i in range(1, len(maps))]
print("fraction of workgroups on the same SIMD as in launch 0:", " ".join(f"{x:.3f}" for x in same))
L.wbc_debug_set_wave_timing(None)
# how well does the previous launch predict a wave's cost?  classes of the PREVIOUS launch: reset / in contact / airborne
L.wbc_debug_set_wave_timing(buf.data_ptr())
prev = None; rows = []
for rep in range(40):
    g.step(acts[rep % 8]); torch.cuda.synchronize()
    t = buf.cpu().numpy().reshape(n, 6)
    fl = t[:, 2]; dur = (t[:, 1] - t[:, 0]).astype(float); rst = (fl & 1) == 1; ncon = (fl >> 8) & 255
    hint = ((fl >> 1) & 1) == 1
    if prev is not None:
        prst, pncon, phint = prev
        for nm, sel in (("hint set, no reset", phint & ~prst), ("hint clear or reset", ~(phint & ~prst))):
            if sel.sum(): rows.append((nm, sel.mean(), dur[sel].mean(), np.percentile(dur[sel], 90), ((ncon[sel] > 0) | rst[sel]).mean(), rst[sel].mean()))
        for nm, sel in (("prev reset", prst), ("prev contact, no reset", (pncon > 0) & ~prst), ("prev airborne, no reset", (pncon == 0) & ~prst)):
            if sel.sum(): rows.append((nm, sel.mean(), dur[sel].mean(), np.percentile(dur[sel], 90), ((ncon[sel] > 0) | rst[sel]).mean(), rst[sel].mean()))
    prev = (rst, ncon, hint)
for nm in ("prev reset", "prev contact, no reset", "prev airborne, no reset", "hint set, no reset", "hint clear or reset"):
    r = np.array([x[1:] for x in rows if x[0] == nm])
    print(f"{nm:26s}: share {r[:, 0].mean():.3f}; duration mean {r[:, 1].mean():.0f} p90 {r[:, 2].mean():.0f}; heavy now {r[:, 3].mean():.3f}; resets now {r[:, 4].mean():.3f}")
L.wbc_debug_set_wave_timing(None)
