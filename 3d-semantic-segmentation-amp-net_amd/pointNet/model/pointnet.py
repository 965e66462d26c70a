"""Drop-in for the reference's pointNet/model/pointnet.py (1024-d baseline PointNet, convolutions with bias):
TransformationNet :6-44, BasePointNet :47-97, ClassificationPointNet :100-125, SegmentationPointNet :128-154.  Same constructor arguments, state_dict
keys and SegmentationPointNet.forward contract; the eval forward runs through ampnet_pointnet_seg_fwd_f32 (variant 0).
The reference builds these modules on the CPU and moves them with .cuda(); here `device` defaults to 'cuda'."""
from . import _baseline as _B

_G, _F1, _F2 = 1024, 512, 256


class TransformationNet(_B.TnetHolder):
    def __init__(self, input_dim, output_dim, device='cuda'):
        super().__init__(input_dim, output_dim, _G, _F1, _F2, True, device)


class BasePointNet(_B.BaseHolder):
    def __init__(self, point_dimension, return_local_features=False, dataset='', device='cuda'):
        if point_dimension != 3:
            raise NotImplementedError("pointnet.py slices x[:, :, :3] for the input T-Net (pointnet.py:71): point_dimension must be 3")
        super().__init__(point_dimension, return_local_features, _G, _F1, _F2, True, device)
        self.dataset = dataset


class SegmentationPointNet(_B.SegHolder):
    VARIANT = 0

    def __init__(self, num_classes, point_dimension=3, device='cuda'):
        super().__init__()
        self.base_pointnet = BasePointNet(return_local_features=True, point_dimension=point_dimension, device=device)
        self._init_head(num_classes, _G, 512, 256, 128, device)


class ClassificationPointNet(_B.ClsHolder):
    """pointnet.py:100-125: global feature -> fc 1024 -> 512 -> 256 (BatchNorm + ReLU) -> Dropout -> log_softmax(fc 256 -> num_classes)."""
    VARIANT = 0

    def __init__(self, num_classes, dropout=0.3, point_dimension=3, dataset='', device='cuda'):
        super().__init__()
        self.dataset = dataset
        self.base_pointnet = BasePointNet(return_local_features=False, point_dimension=point_dimension, dataset=dataset, device=device)
        self._init_head(num_classes, dropout, _G, 512, 256, True, device)
