"""Dev tool: where does the HOST spend its time in GraphedTrainStep.step()?  (the synchronous-loop rate of bench.py is the replay
time plus whatever the host does between two replays)"""
import sys, os, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import uda_poseestimation_amd.lib.models as models
from uda_poseestimation_amd import synthetic
from uda_poseestimation_amd.engine import GraphedTrainStep, MeanTeacherTrainer
dev = torch.device("cuda:0")
torch.manual_seed(0)
stu = models.pose_resnet101(num_keypoints=16, pretrained_backbone=False).to(dev)
tea = models.pose_resnet101(num_keypoints=16, pretrained_backbone=False).to(dev)
tr = MeanTeacherTrainer(stu, tea, precision="bf16")
b = synthetic.mean_teacher_batch(32, seed=0)
g = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()}
args = (g["x_s"], g["label_s"], g["weight_s"], g["x_t_stu"], g["x_t_tea"], g["aug_param_stu"], g["aug_param_tea"])
gs = GraphedTrainStep(tr, *args)
for _ in range(20):
    gs.step(*args)
torch.cuda.synchronize()
# host time of step() with an idle GPU in front (synchronised loop)
ts = []
for _ in range(50):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    gs.step(*args)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    ts.append((t1 - t0, t2 - t0))
ts.sort()
print(f"synchronised loop: host time in step() median {ts[25][0] * 1e3:.3f} ms, step() + sync median {sorted(t[1] for t in ts)[25] * 1e3:.3f} ms")
t0 = time.perf_counter()
for _ in range(50):
    gs.step(*args)
torch.cuda.synchronize()
print(f"back to back: {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms per step")
pr = cProfile.Profile()
pr.enable()
for _ in range(50):
    torch.cuda.synchronize()
    gs.step(*args)
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(22)
print(s.getvalue()[:3500])
