"""Drop-in for the reference's pointNet/model/light_pointnet_256.py (256-d baseline PointNet, no conv / fc bias, input
T-Net on x and y only): TransformationNet :7-45, BasePointNet :48-96, ClassificationPointNet :100-125, SegmentationPointNet :128-153.  As committed, the
reference model only runs with point_dimension=2 (it slices x[:, :, :2] at :71); the same restriction holds
here.  Eval forward through ampnet_pointnet_seg_fwd_f32 (variant 1)."""
from . import _baseline as _B

_G, _F1, _F2 = 256, 256, 128


class TransformationNet(_B.TnetHolder):
    def __init__(self, input_dim, output_dim, device='cuda'):
        super().__init__(input_dim, output_dim, _G, _F1, _F2, False, device)


class BasePointNet(_B.BaseHolder):
    def __init__(self, point_dimension, return_local_features=False, device='cuda'):
        if point_dimension != 2:
            raise NotImplementedError("light_pointnet_256.py slices x[:, :, :2] for the input T-Net (:71): point_dimension must be 2 "
                                      "(point_dimension=3 raises in the reference too)")
        super().__init__(point_dimension, return_local_features, _G, _F1, _F2, False, device)


class SegmentationPointNet(_B.SegHolder):
    VARIANT = 1

    def __init__(self, num_classes, point_dimension=3, device='cuda'):
        super().__init__()
        self.base_pointnet = BasePointNet(return_local_features=True, point_dimension=point_dimension, device=device)
        self._init_head(num_classes, _G, 256, 128, 64, device)


class ClassificationPointNet(_B.ClsHolder):
    """light_pointnet_256.py:100-125: global feature -> fc 256 -> 128 -> 64 (no bias; BatchNorm + ReLU) -> Dropout -> log_softmax(fc 64 ->
    num_classes).  The reference's default point_dimension=3 cannot run (its BasePointNet slices x[:, :, :2], :71): pass point_dimension=2."""
    VARIANT = 1

    def __init__(self, num_classes, dropout=0.3, point_dimension=3, dataset='', device='cuda'):
        super().__init__()
        self.dataset = dataset
        self.base_pointnet = BasePointNet(return_local_features=False, point_dimension=point_dimension, device=device)
        self._init_head(num_classes, dropout, _G, 128, 64, False, device)
