import sys, time, torch
sys.path.insert(0, '/root/repo')
from cmoop_audio_processing_amd import frontend
wav = torch.randn((30000, 16000), device='cuda')
for _ in range(2): frontend.log_mel(wav[:256])
torch.cuda.synchronize()
best=1e9
for _ in range(5):
    t=time.perf_counter(); out=frontend.log_mel(wav); best=min(best,time.perf_counter()-t)
print("frontend 30000 clips best ms", best*1e3, "GB/s", (wav.numel()*4+out.numel()*4)/best/1e9)
