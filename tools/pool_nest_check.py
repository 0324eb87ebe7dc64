import sys; sys.path.insert(0,'.')
import torch
from biahub_amd.device import volume_pool, empty
from biahub_amd.deskew import fast_deskew_zyx
dev=torch.device('cuda',0)
with volume_pool(dev):
    a = empty((64,256,256), torch.float32, dev)
    with volume_pool(dev):
        b = torch.empty((300, 1024, 1024), dtype=torch.float32, device=dev)
    out = fast_deskew_zyx(a.normal_(), 36.17, 0.371, True, 3, "mean")
print('nested ok', out.shape)
