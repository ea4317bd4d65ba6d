"""Round 4: what clock and board power the chip sustains under each kernel family (rocm-smi sampled every 0.25 s while one
shape is launched back to back for ~4 s).  The dense-MFMA peak of MI355X_MICROARCH.md (2.5 PFLOP/s f16) is quoted at 2.4 GHz."""
import json, os, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "guidedvd-3dgs_amd"))
import torch
import torch.nn.functional as F
from lvdm_amd import gemm, ops, conv

dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
samples, stop = [], [False]


def sampler():
    while not stop[0]:
        try:
            out = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5).stdout
            d = json.loads(out)
            card = d[sorted(d)[0]]
            rec = {}
            for k, v in card.items():
                kl = k.lower()
                if "sclk" in kl and "level" not in kl or ("sclk" in kl and "(" in str(v)):
                    rec["sclk"] = str(v)
                if "power" in kl and "w" in kl:
                    rec["power"] = str(v)
            rec["raw"] = {k: v for k, v in card.items() if "sclk" in k.lower() or "power" in k.lower()}
            samples.append((time.time(), rec))
        except Exception as e:  # noqa: BLE001
            samples.append((time.time(), {"err": str(e)}))
        time.sleep(0.25)


def run(name, fn, flops, secs=4.0):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.time()
    n = 0
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    while time.time() - t0 < secs:
        for _ in range(20):
            fn()
        n += 20
        torch.cuda.synchronize()
    e.record()
    torch.cuda.synchronize()
    t1 = time.time()
    ms = a.elapsed_time(e) / n
    mine = [r for (t, r) in samples if t0 + 1.0 <= t <= t1]
    print(f"{name:44s} {ms * 1e3:9.1f} us  {flops / ms / 1e9:7.0f} TFLOP/s   samples: " + " | ".join(json.dumps(r.get("raw", r)) for r in mine[-3:]), flush=True)


th = threading.Thread(target=sampler, daemon=True)
th.start()
time.sleep(1.5)
print("idle: " + " | ".join(json.dumps(r.get("raw", r)) for (_, r) in samples[-2:]), flush=True)
for (M, N, K) in [(230400, 2560, 320), (57600, 5120, 640), (14400, 10240, 1280)]:
    x = torch.randn(M, K, device=dev, generator=g).half()
    w = (torch.randn(N, K, device=dev, generator=g) * K ** -0.5).half()
    b = torch.randn(N, device=dev, generator=g)
    bh = b.half()
    run(f"gemm {M}x{N}x{K} ours", lambda: gemm.gemm_nt(x, w, bias=b), 2.0 * M * N * K)
    run(f"gemm {M}x{N}x{K} hipBLASLt", lambda: F.linear(x, w, bh), 2.0 * M * N * K)
# level-0 self-attention (25 frames x 5 heads, 9216 tokens) and two convolutions (U-Net 72 x 128 640 -> 640, VAE 576 x 1024 128 -> 128)
q, k, v = (torch.randn(25, 9216, 320, device=dev, generator=g).half() for _ in range(3))
run("attention fwd 25x5 heads N=9216 d=64", lambda: ops._hip_attention_fwd(q, k, v, 5, False, want_lse=True), 4.0 * 25 * 5 * 9216 * 9216 * 64)
del q, k, v
for (N, H, W, Cin, Cout) in [(25, 72, 128, 640, 640), (1, 576, 1024, 128, 128)]:
    xc = torch.randn(N, H, W, Cin, device=dev, generator=g).half()
    m = torch.nn.Conv2d(Cin, Cout, 3, padding=1).to(dev).half().requires_grad_(False)
    with torch.no_grad():
        run(f"conv3x3 N={N} {H}x{W} {Cin}->{Cout}", lambda: conv.fused_conv(xc, m), 2.0 * N * H * W * Cin * Cout * 9)
    del xc, m
xm = torch.randn(8 * 1024 * 1024, device=dev, generator=g).half()
run("copy 16 MB x2 (HBM stream)", lambda: xm.clone(), 0.0)
stop[0] = True
