#!/usr/bin/env python3
"""Development aid: time one convolution shape under forced launch plans (CDC_PLAN=MB,NPW,KC).
usage: gpu_conv_tune.py B Cin H W Cout k stride fusedLN(0/1) [plan ...]   plan = MB,NPW,KC"""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def child(args):
    from cdc_compression_amd import synth
    from cdc_compression_amd.ops import Ops
    B, Ci, H, W, Co, k, s, ln = [int(a) for a in args]
    G = Ops(0)
    rng = np.random.default_rng(0)
    x = rng.standard_normal((B, Ci, H, W)).astype(np.float32)
    w = (rng.standard_normal((Co, Ci, k, k)) / np.sqrt(Ci * k * k)).astype(np.float32)
    b = rng.standard_normal(Co).astype(np.float32) * 0.1
    kw = {}
    if ln:
        kw = dict(ln_g=np.ones(Co, np.float32), ln_b=np.zeros(Co, np.float32), relu=True)
    if os.environ.get("TUNE_RESID"):
        kw["resid"] = rng.standard_normal((B, Co, H // s, W // s)).astype(np.float32)
    G.conv2d(x, w, b, s, k // 2, **kw)
    ts = []
    for _ in range(5):
        G.prof(True)
        G.conv2d(x, w, b, s, k // 2, **kw)
        ms, n, fl = G.prof_total_ms()
        ts.append((ms, n, fl))
    ms, n, fl = sorted(ts)[len(ts) // 2]
    print(f"plan {os.environ.get('CDC_PLAN', 'auto'):>8}: {ms:8.3f} ms  {fl / ms / 1e9:7.1f} TF  ({n} launches)", flush=True)

if __name__ == "__main__":
    if os.environ.get("TUNE_CHILD"):
        child(sys.argv[1:9])
    else:
        shape, plans = sys.argv[1:9], sys.argv[9:] or ["auto"]
        print("shape B,Cin,H,W,Cout,k,s,LN =", " ".join(shape), flush=True)
        for p in plans:
            env = dict(os.environ, TUNE_CHILD="1", CDC_DEBUG_PLAN="1", CDC_DEV="1")   # CDC_PLAN is a development switch
            if p != "auto":
                env["CDC_PLAN"] = p
            r = subprocess.run([sys.executable, __file__] + shape, env=env, capture_output=True, text=True)
            plan_line = [l for l in r.stderr.splitlines() if l.startswith("[plan]")]
            print((plan_line[0][60:] if plan_line else "   (no plan)") + "\n   " + (r.stdout.strip() or r.stderr.strip()[-300:]), flush=True)
