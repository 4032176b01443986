"""Dev tool: img/s of the EAGER step on configs[1] (every kernel launched from the host): three streams (engine.train_step) and one
stream (what a reference-style loop calling student(x_s); student(x_t); teacher(x_t) in turn gets from the same kernels)."""
import sys, time
import torch
sys.path.insert(0, ".")
import uda_poseestimation_amd.lib.models as models
from uda_poseestimation_amd import synthetic
from uda_poseestimation_amd.engine import MeanTeacherTrainer
N, K, S = 32, 16, 256
dev = torch.device("cuda:0")
stu = models.pose_resnet101(num_keypoints=K, pretrained_backbone=False).to(dev)
tea = models.pose_resnet101(num_keypoints=K, pretrained_backbone=False).to(dev)
tr = MeanTeacherTrainer(stu, tea, image_size=S, heatmap_size=S // 4)
b = synthetic.mean_teacher_batch(N, num_keypoints=K, image_size=S, heatmap_size=S // 4, seed=0)
g = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()}
args = (g["x_s"], g["label_s"], g["weight_s"], g["x_t_stu"], g["x_t_tea"], g["aug_param_stu"], g["aug_param_tea"])
for conc in (True, False, True, False):
    tr.concurrent = conc
    for _ in range(40):
        tr.train_step(*args)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(40):
        tr.train_step(*args)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 40 * 1e3
    print(f"eager, {'three streams' if conc else 'one stream   '}: {ms:.2f} ms/step = {N / ms * 1e3:.0f} img/s", flush=True)
