import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.vispeech_oracle import rel_attention
from vispeech_amd import config as vcfg
from vispeech_amd.models import SynthesizerTrn
from vispeech_amd.schema import ModelDims
from vispeech_amd.synth import synth_state_dict
dims = ModelDims(); sd = synth_state_dict(dims, seed=1234, infer_only=True)
a, kw = vcfg.synthesizer_args(vcfg.default_hparams())
net = SynthesizerTrn(*a, **kw).eval(); net.load_state_dict(sd)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 300
lens = [T, max(T // 2 - 27, 1)]
r = np.random.Generator(np.random.PCG64(T)); B, H = 2, 192
qkv = r.standard_normal((B, 3 * H, T)).astype(np.float32)
la = np.array(lens, dtype=np.int64)
prefix, layer, which = "enc_p.encoder", 1, 0
ek = torch.from_numpy(sd[f"{prefix}.attn_layers.{layer}.emb_rel_k"]); ev = torch.from_numpy(sd[f"{prefix}.attn_layers.{layer}.emb_rel_v"])
t = torch.from_numpy(qkv); mask = (torch.arange(T)[None, :] < torch.from_numpy(la)[:, None])
ref = rel_attention(t[:, :H], t[:, H:2*H], t[:, 2*H:], ek, ev, mask, 2, 4).numpy()
out = net._engine.attention(which, layer, qkv, la).cpu().numpy()
for b, n in enumerate(lens):
    e = np.abs(out[b, :, :n] - ref[b, :, :n]); sc = np.abs(ref[b, :, :n]).max()
    print("b", b, "n", n, "max rel", e.max() / sc, "argmax (d, i)", np.unravel_index(e.argmax(), e.shape))
    print("  per head:", [float(e[h*96:(h+1)*96].max() / sc) for h in range(2)])
    print("  per query block of 16:", " ".join(f"{e[:, i:i+16].max()/sc:.1e}" for i in range(0, n, 16)))
    print("  per d mod 16:", " ".join(f"{e[np.arange(192) % 16 == m].max()/sc:.1e}" for m in range(16)))
    # without the relative-value term?  error vs mean |v| etc
print("---- row detail, b=0")
b = 0; n = lens[0]
e = np.abs(out[b, :, :n] - ref[b, :, :n]); sc = np.abs(ref[b, :, :n]).max()
ekn = ek.numpy()[0]
for hd in range(2):
    q = qkv[b, hd*96:(hd+1)*96].T / np.sqrt(96); k = qkv[b, H+hd*96:H+(hd+1)*96].T
    S = q @ k.T
    rl = q @ ekn.T
    for i in range(n):
        for rr in range(9):
            j = i + rr - 4
            if 0 <= j < n: S[i, j] += rl[i, rr]
    er = e[hd*96:(hd+1)*96].max(axis=0) / sc
    bad = np.where(er > 3e-6)[0]
    print("head", hd, "bad rows", bad.tolist())
    for i in bad[:12]:
        order = np.argsort(-S[i])
        run_max_tile = [int(np.argmax(S[i, :32*(t+1)])) // 32 for t in range((n+31)//32)]
        print(f"   row {i}: err {er[i]:.1e} argmax key {order[0]} (tile {order[0]//32}) smax {S[i,order[0]]:.2f} 2nd {S[i,order[1]]:.2f} running-argmax tiles {run_max_tile}")
print("---- error direction vs Ev rows / V rows")
evn = ev.numpy()[0]   # [9, 96]
for hd, rows in ((0, [90]), (1, [208, 265])):
    v = qkv[b, 2*H+hd*96:2*H+(hd+1)*96]   # [96, T]
    for i in rows:
        d = (out[b, hd*96:(hd+1)*96, i] - ref[b, hd*96:(hd+1)*96, i]).astype(np.float64)
        print(f"row {i}: |d| {np.linalg.norm(d):.2e}")
        for rr in range(9):
            c = d @ evn[rr] / (evn[rr] @ evn[rr]); res = np.linalg.norm(d - c * evn[rr]) / np.linalg.norm(d)
            if res < 0.5: print(f"    ~ {c:.3e} * Ev[{rr}]  (residual {res:.2f})")
        for j in range(max(0, i-40), min(n, i+40)):
            c = d @ v[:, j] / (v[:, j] @ v[:, j]); res = np.linalg.norm(d - c * v[:, j]) / np.linalg.norm(d)
            if res < 0.5: print(f"    ~ {c:.3e} * V[{j}]  (residual {res:.2f})")
print("---- vs own output / any V column")
for hd, rows in ((0, [90]), (1, [208, 265])):
    v = qkv[b, 2*H+hd*96:2*H+(hd+1)*96]
    for i in rows:
        d = (out[b, hd*96:(hd+1)*96, i] - ref[b, hd*96:(hd+1)*96, i]).astype(np.float64)
        o_ = ref[b, hd*96:(hd+1)*96, i].astype(np.float64)
        c = d @ o_ / (o_ @ o_); print(f"row {i}: vs own output c={c:.3e} residual {np.linalg.norm(d - c*o_)/np.linalg.norm(d):.2f}")
        best = []
        for j in range(n):
            c = d @ v[:, j] / (v[:, j] @ v[:, j]); res = np.linalg.norm(d - c * v[:, j]) / np.linalg.norm(d)
            best.append((res, j, c))
        best.sort()
        print("    best V columns:", [(round(r_, 2), j, f"{c:.2e}") for r_, j, c in best[:3]])
print("---- hypothesis: one q element of the row is off")
for hd, rows in ((0, [90]), (1, [208, 265])):
    q = qkv[b, hd*96:(hd+1)*96].T.astype(np.float64) / np.sqrt(96); k = qkv[b, H+hd*96:H+(hd+1)*96].T.astype(np.float64)
    v = qkv[b, 2*H+hd*96:2*H+(hd+1)*96].astype(np.float64)
    S = q @ k.T
    rl = q @ ekn.T.astype(np.float64)
    for i in range(n):
        for rr in range(9):
            j = i + rr - 4
            if 0 <= j < n: S[i, j] += rl[i, rr]
    for i in rows:
        p = np.exp(S[i] - S[i].max()); p /= p.sum()
        d = (out[b, hd*96:(hd+1)*96, i] - ref[b, hd*96:(hd+1)*96, i]).astype(np.float64)
        best = []
        for d0 in range(96):
            ds = k[:, d0]
            dp = p * (ds - (p * ds).sum())
            g = v @ dp
            c = d @ g / (g @ g); res = np.linalg.norm(d - c * g) / np.linalg.norm(d)
            best.append((res, d0, c))
        best.sort()
        print(f"row {i}: best q-element explanations:", [(round(r_, 3), d0, f"{c:.2e}", f"q={q[i,d0]*np.sqrt(96):.4f}") for r_, d0, c in best[:3]])
