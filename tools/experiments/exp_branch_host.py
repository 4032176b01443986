"""Dev tool: host-side enqueue time against device time of the branch-graph step (engine.GraphedTrainStep(branch_graphs=True))."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import uda_poseestimation_amd
import torch
from uda_poseestimation_amd import synthetic
from uda_poseestimation_amd.engine import GraphedTrainStep, MeanTeacherTrainer
import uda_poseestimation_amd.lib.models as models
dev = torch.device("cuda:0")
torch.manual_seed(0)
stu = models.pose_resnet101(16, pretrained_backbone=False).to(dev)
tea = models.pose_resnet101(16, pretrained_backbone=False).to(dev)
tr = MeanTeacherTrainer(stu, tea, precision="bf16")
b = synthetic.mean_teacher_batch(32, seed=0)
g = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()}
gs = GraphedTrainStep(tr, g["x_s"], g["label_s"], g["weight_s"], g["x_t_stu"], g["x_t_tea"], g["aug_param_stu"], g["aug_param_tea"],
                      branch_graphs=(os.environ.get("BR", "1") == "1"))
def step():
    return gs.step(None, None, None, None, None, g["aug_param_stu"], g["aug_param_tea"])
for _ in range(30):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')} branch={gs.branch}: host enqueue {(t1 - t0) / 20 * 1e3:.2f} ms per step, wall {(t2 - t0) / 20 * 1e3:.2f} ms per step", flush=True)
if gs.branch:
    # each graph alone, back to back on its stream
    for name, gr in gs._bg.items():
        torch.cuda.synchronize()
        st = gs._bs["main"]
        with torch.cuda.stream(st):
            t0 = time.perf_counter()
            for _ in range(10):
                gr.replay()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
        print(f"  graph {name:8s}: host {(t1 - t0) / 10 * 1e3:6.3f} ms per replay, wall {(t2 - t0) / 10 * 1e3:6.3f} ms", flush=True)

if gs.branch and os.environ.get("TL", "1") == "1":
    # device timeline of one step: timing events around every graph replay on its stream
    import types
    marks = []
    orig = {n: gr.replay for n, gr in gs._bg.items()}
    class Wrap:
        def __init__(self, name, gr): self.name, self.gr = name, gr
        def replay(self):
            a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); self.gr.replay(); b_.record()
            marks.append((self.name, a, b_))
    real = dict(gs._bg)
    gs._bg = {n: Wrap(n, gr) for n, gr in real.items()}
    for it in range(3):
        marks.clear()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e0.record()
        th0 = time.perf_counter()
        step()
        th1 = time.perf_counter()
        torch.cuda.synchronize()
    print(f"  one step: host {(th1 - th0) * 1e3:.2f} ms; device timeline (ms from the step's start):")
    for name, a, b_ in marks:
        print(f"    {name:8s} {e0.elapsed_time(a):7.2f} -> {e0.elapsed_time(b_):7.2f}")
