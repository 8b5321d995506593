import cProfile, pstats, sys, io, torch
sys.path.insert(0, '/root/repo')
import bench
from vqa_amd import ops
dev = torch.device('cuda:0')
fus, gsm, opt, main_opt, sets = bench.build_job(dev, 1024, torch.bfloat16, 0)
step = bench.make_step(fus, gsm, opt, main_opt, sets)
for i in range(8): step(i)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for i in range(40): step(i)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(28); print(s.getvalue()[:6000])
