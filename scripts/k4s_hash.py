"""K4s on four shapes, soft and hard WPMI: sha256 of the outputs against the values recorded in profiles/r03_k4s_ring.txt --
this round's restructurings of the kernel (metadata array, LDS ring, software pipeline) all left the bits where they were."""
import torch, hashlib, sys
sys.path.insert(0, '.')
import mammo_clip_dissect_amd
from mammo_clip_dissect_amd import core
dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(5)
want = {(3000, 1000, 300, 100): ("31ce15d9bec62375", "f1d32bfd621748fc"), (777, 333, 50, 37): ("ff72fe47a268d0e9", "ff59896cf73b098d"),
        (5000, 2560, 129, 16): ("61d708126ceb9e08", "45af029f8cd8305e"), (64, 128, 7, 3): ("865d7af3a158e65b", "678732b102e49c82")}
ok = True
for (N, C, U, K) in [(3000, 1000, 300, 100), (777, 333, 50, 37), (5000, 2560, 129, 16), (64, 128, 7, 3)]:
    I = core.normalize_rows(torch.randn(N, 512, device=dev, generator=g)); T = core.normalize_rows(torch.randn(C, 512, device=dev, generator=g))
    E, rinv = core.embed_gemm_exp(I, T, 10.0)
    idx = torch.stack([torch.randperm(N, device=dev, generator=g)[:K] for _ in range(U)]).to(torch.int32)
    p = torch.linspace(0.998, 0.97, K, device=dev)
    outs = core.wpmi_score_bf16(E, rinv, idx, p, 1e-9, soft=True)
    outh = core.wpmi_score_bf16(E, rinv, idx, None, 1e-9, soft=False)
    got = (hashlib.sha256(outs.cpu().numpy().tobytes()).hexdigest()[:16], hashlib.sha256(outh.cpu().numpy().tobytes()).hexdigest()[:16])
    print((N, C, U, K), got, "same as the round's earlier kernel" if got == want[(N, C, U, K)] else "DIFFERENT")
    ok &= got == want[(N, C, U, K)]
print("bit-identical" if ok else "MISMATCH")
