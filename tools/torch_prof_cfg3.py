"""torch.profiler view of one cfg3 forward: which aten ops the stand-alone elementwise / copy kernels belong to."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
os.environ.setdefault("GKG_RELPOS_DEVICE", "cuda")
from gkgnet_amd import layers
from gkgnet_amd.backbone import GKGNet
layers.norm_cfg["type"] = "BN"
net = GKGNet(choice="s", k=9, k_label_gcn=9, n_classes=80, size=576).cuda().eval()
img = torch.randn(32, 3, 576, 576, device="cuda")
for _ in range(2):
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        net(img)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False, record_shapes=True) as prof:
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        net(img)
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_cuda_time_total", row_limit=45, max_name_column_width=40, max_shapes_column_width=60))
