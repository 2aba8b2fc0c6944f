"""Per-wave durations of ONE wbc_step_kernel launch (timing build: tools/build_variant.py timing -DWBC_STEP_TIMING): how far is the
launch (= its slowest SIMD) from the mean wave, and what do the slow waves have in common?  usage: python tools/wave_spread.py [N]"""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-whole-body-control_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
_w = os.path.join(ROOT, "deep-whole-body-control_amd", "wbc_amd", "libwbc_amd_wavetiming.so")      # tools/build_variant.py wavetiming -DWBC_WAVE_TIMING
os.environ["WBC_AMD_LIB"] = _w if os.path.exists(_w) else os.path.join(ROOT, "deep-whole-body-control_amd", "wbc_amd", "libwbc_amd_timing.so")
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
for rep in range(3):
    g.step(acts[rep]); torch.cuda.synchronize()
    t = buf.cpu().numpy().reshape(n, 6)
    t0, t1, fl = t[:, 0], t[:, 1], t[:, 2]
    dur = (t1 - t0).astype(np.float64); rst = (fl & 1) == 1
    ncon = (fl >> 8) & 255; dmax = (fl >> 16) & 255; nself = np.zeros_like(ncon)      # of the last substep
    span = t1.max() - t0.min()
    print(f"launch {rep}: span {span} cycles; wave duration mean {dur.mean():.0f} p50 {np.median(dur):.0f} p90 {np.percentile(dur, 90):.0f} p99 {np.percentile(dur, 99):.0f} max {dur.max():.0f}; "
          f"start spread {t0.max() - t0.min()}; resets {rst.mean():.3f}: mean duration reset {dur[rst].mean():.0f} / no reset {dur[~rst].mean():.0f}")
    sub, post, rs, obs = (t[:, 3] - t0).astype(float), (t[:, 4] - t[:, 3]).astype(float), (t[:, 5] - t[:, 4]).astype(float), (t1 - t[:, 5]).astype(float)
    for nm, x in (("load + 4 substeps", sub), ("rigid bodies + task logic + rewards", post), ("reset", rs), ("observe + store", obs)):
        print(f"   {nm:38s} no reset: mean {x[~rst].mean():8.0f} p90 {np.percentile(x[~rst], 90):8.0f} max {x[~rst].max():8.0f} | reset: mean {x[rst].mean():8.0f} p90 {np.percentile(x[rst], 90):8.0f} max {x[rst].max():8.0f}")
    for k in sorted(set(ncon.tolist())):
        sel = ncon == k
        if sel.sum() >= 20: print(f"   active contacts {k:2d}: {sel.sum():5d} waves, substeps mean {sub[sel].mean():8.0f} (deepest level mean {dmax[sel].mean():.1f}, self-collision pairs {nself[sel].mean():.2f})")
    hw = (fl >> 32) & 0xFFFF; xcc = (fl >> 24) & 15
    simd = ((hw >> 4) & 3) | (((hw >> 8) & 15) << 2) | (((hw >> 12) & 1) << 6) | (((hw >> 13) & 7) << 7) | (xcc << 10)     # simd, cu, sh, se, xcc
    heavy = (ncon > 0) | rst
    ids, inv = np.unique(simd, return_inverse=True)
    nh = np.bincount(inv, weights=heavy.astype(float)); nw = np.bincount(inv)
    # a SIMD's busy time: from its first wave's start to its last wave's end (stamps of one XCD share a clock)
    first = np.full(len(ids), np.iinfo(np.int64).max); last = np.zeros(len(ids), dtype=np.int64)
    np.minimum.at(first, inv, t0); np.maximum.at(last, inv, t1)
    busy = (last - first).astype(float)
    print(f"   {len(ids)} SIMDs seen, waves per SIMD min {nw.min()} max {nw.max()}; heavy waves (in contact or resetting) {heavy.mean():.3f}")
    for k in range(0, int(nh.max()) + 1):
        sel = nh == k
        if sel.sum(): print(f"   SIMDs with {k} heavy waves: {sel.sum():4d}, busy cycles mean {busy[sel].mean():8.0f} max {busy[sel].max():8.0f}")
    late = dur > np.percentile(dur, 99)
    print(f"   the slowest 1 % of waves, phase means: substeps {sub[late].mean():.0f} post {post[late].mean():.0f} reset {rs[late].mean():.0f} observe {obs[late].mean():.0f}; active contacts {ncon[late].mean():.1f}")
    print(f"   the slowest 1 % of waves: reset share {rst[late].mean():.2f}, mean duration {dur[late].mean():.0f}, mean start offset {(t0[late] - t0.min()).mean():.0f}")
L.wbc_debug_set_wave_timing(None)
