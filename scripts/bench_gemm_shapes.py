"""Time the dense NT contractions of a training step (shapes from profiles/r3_gemm_trace.txt) under the kernel the
dispatcher picks; run once per DFOLD_GEMM_VARIANT (256 default / 2560 no 256x320 kernel / 128 only 128x128) to compare."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dynamicpdb_amd import ops
from dynamicpdb_amd.ops import gemm, rows_plain
dev = torch.device("cuda:0")
shapes = [(65536, 4096, 256, "bf16"), (65536, 3072, 256, "bf16"), (65536, 2048, 256, "bf16"), (65536, 256, 4096, "bf16"),
          (65536, 256, 3072, "f32"), (65536, 256, 2048, "bf16"), (65536, 1280, 1280, "bf16"), (65536, 256, 256, "f32"),
          (65536, 256, 256, "bf16"), (65536, 480, 256, "f32"), (65536, 192, 256, "f32"), (65536, 256, 480, "bf16"), (524288, 128, 128, "bf16")]
for M, N, K, dt in shapes:
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    B = torch.randn(N, K, device=dev).to(torch.bfloat16)
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16 if dt == "bf16" else torch.float32)
    f = lambda: gemm(A, B, C, M, N, K, a_rows=rows_plain(K), c_rows=rows_plain(N), ldb=K)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10
    by = (M * K + N * K) * 2 + M * N * C.element_size()
    print(f"variant {os.environ.get('DFOLD_GEMM_VARIANT','256'):>5s}  {M:7d} x {N:5d} x {K:5d} {dt:4s}: {t*1e3:8.1f} us  {2.0*M*N*K/t/1e9:8.1f} TFLOP/s  {by/t/1e6:7.1f} GB/s", flush=True)
