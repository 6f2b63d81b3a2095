for v in ${VARIANTS:-cur v1}; do
  if [ "$v" != cur ]; then export LEFTREFILL_LIB_PATH=leftrefill_amd/lib/variants/libleftrefill_hip_$v.so; else unset LEFTREFILL_LIB_PATH; fi
  echo "== variant [$v]"
  for sh in "65536 320 2880 9 256 320" "16384 640 5760 9 256 160"; do
    python - $sh <<'PY'
import sys, torch
sys.path.insert(0, '.')
from leftrefill_amd import ops
import tools.bench_gemm as bg
M,N,K,taps,tm,tn = map(int, sys.argv[1:7])
import types
dev = torch.device("cuda:0")
C=K//taps; B=8; HW=M//B; W=int((HW*2)**0.5); H=HW//W
x=torch.randn(M,C,device=dev).half(); w=(torch.randn(N,K,device=dev)/K**0.5).half(); b=torch.randn(N,device=dev)
out=torch.empty(M,N,device=dev,dtype=torch.float16)
for pipe in (0,8):
    f=lambda: ops.gemm_conv(x,w,B=B,H=H,W=W,taps=taps,bias=b,out=out,tile_m=tm,tile_n=tn,splits=1,pipe=pipe)
    for _ in range(3): f()
    torch.cuda.synchronize()
    best=1e9
    for r in range(3):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); e1.synchronize()
        best=min(best,1e3*e0.elapsed_time(e1)/20)
    print(f"M={M} N={N} K={K} tile {tm}x{tn} pipe {pipe}: {best:7.1f} us {2.0*M*N*K/best/1e6:7.1f} TF")
PY
  done
done
