"""Dev tool: full-depth nets at other image sizes / batches (eager + captured step)."""
import sys
sys.path.insert(0, '.')
import torch
from uda_poseestimation_amd import synthetic
from uda_poseestimation_amd.engine import GraphedTrainStep, MeanTeacherTrainer
import uda_poseestimation_amd.lib.models as models
for arch, K, N, S in (("pose_resnet50", 16, 3, 224), ("pose_resnet50", 21, 7, 320), ("pose_resnet101", 14, 2, 352), ("pose_resnet101", 16, 48, 256)):
    torch.manual_seed(0)
    stu = models.__dict__[arch](num_keypoints=K, pretrained_backbone=False).cuda()
    tea = models.__dict__[arch](num_keypoints=K, pretrained_backbone=False).cuda()
    tr = MeanTeacherTrainer(stu, tea, image_size=S, heatmap_size=S // 4)
    b = synthetic.mean_teacher_batch(N, num_keypoints=K, image_size=S, heatmap_size=S // 4, seed=1)
    g = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}
    args = (g["x_s"], g["label_s"], g["weight_s"], g["x_t_stu"], g["x_t_tea"], g["aug_param_stu"], g["aug_param_tea"])
    o = tr.train_step(*args)
    gs = GraphedTrainStep(tr, *args, warmup=1)
    o2 = gs.step(*args)
    torch.cuda.synchronize()
    print(arch, K, N, S, float(o["loss_all"]), float(o2["loss_all"]), f"{torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
    del gs, tr, stu, tea
    torch.cuda.empty_cache()
