"""Loss-head stand-ins for `bench.py --config trainer` (BASELINE.json configs[4], SURVEY.md §8d "neighbour facts").

These are the CALLERS' networks, not the hot path: stock PyTorch-ROCm modules with random weights whose only job is to
hand the HIP generator's backward a dL/dimage of realistic shape and cost.  None is part of any parity claim.

  * `ArcFaceBackbone`   IR-SE-50 at 112x112 -> 512-d embedding, the network behind the reference's id loss
                         (libs/criteria/id_loss.py:11-34, libs/criteria/model_irse.py:9-48: `Backbone(112, 50, 'ir_se', 0.6)`);
                         `IdLoss` = crop [35:223, 32:220] -> AdaptiveAvgPool2d(112) -> 1 - cos (id_loss.py:20-34).
  * `LpipsShaped`       AlexNet-`features`-shaped conv stack, 5 taps, channel-normalised squared feature distance with 1x1
                         "lin" layers (libs/criteria/lpips/lpips.py:28-34, lpips/utils.py:6-12, lpips/networks.py:76-96).  The
                         real LPIPS needs torchvision's pretrained AlexNet plus a URL download -- neither exists offline.
  * `ShapeModelStandIn` DECA replaced by a ResNet-50-shaped CNN on a 224x224 resize -> 236 parameters
                         (100 shape + 50 tex + 50 exp + 6 pose + 3 cam + 27 light, libs/DECA/decalib/utils/config.py:35-40) and
                         a fixed random linear "landmark" head (68x2) standing in for FLAME; DECA itself needs pytorch3d / kornia /
                         its data files (absent).  Returns the dict + angles that `calculate_shapemodel` returns
                         (libs/utilities/generic.py:22-34) so the shift-vector code consumes it unchanged.
"""
import os
import sys

import torch
import torch.nn.functional as F
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from stylegan_directions_face_reenactment_amd.encoder import ResidualUnit, _TRUNK       # noqa: E402  (the IR-SE units)


class ArcFaceBackbone(nn.Module):
    def __init__(self, drop_ratio=0.6):
        super().__init__()
        self.input_layer = nn.Sequential(nn.Conv2d(3, 64, 3, 1, 1, bias=False), nn.BatchNorm2d(64), nn.PReLU(64))
        units, c = [], 64
        for depth, n in _TRUNK[50]:
            for u in range(n):
                units.append(ResidualUnit(c, depth, 2 if u == 0 else 1, True))
                c = depth
        self.body = nn.Sequential(*units)
        self.output_layer = nn.Sequential(nn.BatchNorm2d(512), nn.Dropout(drop_ratio), nn.Flatten(),
                                          nn.Linear(512 * 7 * 7, 512), nn.BatchNorm1d(512, affine=True))

    def forward(self, x):
        x = self.output_layer(self.body(self.input_layer(x)))
        return x / torch.norm(x, 2, 1, True)


class IdLoss(nn.Module):
    def __init__(self):
        super().__init__()
        self.facenet = ArcFaceBackbone().eval()
        self.face_pool = nn.AdaptiveAvgPool2d((112, 112))

    def feats(self, x):
        return self.facenet(self.face_pool(x[:, :, 35:223, 32:220]))

    def forward(self, y_hat, y):
        return (1 - F.cosine_similarity(self.feats(y_hat), self.feats(y).detach(), dim=1, eps=1e-6)).mean()


class LpipsShaped(nn.Module):
    CH = (64, 192, 384, 256, 256)

    def __init__(self):
        super().__init__()
        c = self.CH
        self.slices = nn.ModuleList([
            nn.Sequential(nn.Conv2d(3, c[0], 11, 4, 2), nn.ReLU()),
            nn.Sequential(nn.MaxPool2d(3, 2), nn.Conv2d(c[0], c[1], 5, padding=2), nn.ReLU()),
            nn.Sequential(nn.MaxPool2d(3, 2), nn.Conv2d(c[1], c[2], 3, padding=1), nn.ReLU()),
            nn.Sequential(nn.Conv2d(c[2], c[3], 3, padding=1), nn.ReLU()),
            nn.Sequential(nn.Conv2d(c[3], c[4], 3, padding=1), nn.ReLU())])
        self.lin = nn.ModuleList([nn.Conv2d(ch, 1, 1, bias=False) for ch in c])
        self.register_buffer('mean', torch.tensor([-.030, -.088, -.188]).view(1, 3, 1, 1))
        self.register_buffer('std', torch.tensor([.458, .448, .450]).view(1, 3, 1, 1))

    def feats(self, x):
        x = (x - self.mean) / self.std
        out = []
        for s in self.slices:
            x = s(x)
            out.append(x / (torch.sqrt(torch.sum(x ** 2, dim=1, keepdim=True) + 1e-9) + 1e-10))
        return out

    def forward(self, x, y):
        res = [l((fx - fy) ** 2).mean((2, 3), True) for fx, fy, l in zip(self.feats(x), self.feats(y), self.lin)]
        return torch.sum(torch.cat(res, 0)) / x.shape[0]


class _Bottleneck(nn.Module):
    def __init__(self, cin, mid, stride):
        super().__init__()
        cout = mid * 4
        self.body = nn.Sequential(nn.Conv2d(cin, mid, 1, bias=False), nn.BatchNorm2d(mid), nn.ReLU(),
                                  nn.Conv2d(mid, mid, 3, stride, 1, bias=False), nn.BatchNorm2d(mid), nn.ReLU(),
                                  nn.Conv2d(mid, cout, 1, bias=False), nn.BatchNorm2d(cout))
        self.short = None if (cin == cout and stride == 1) else \
            nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        return F.relu(self.body(x) + (x if self.short is None else self.short(x)))


class ShapeModelStandIn(nn.Module):
    N_PARAM = 100 + 50 + 50 + 6 + 3 + 27

    def __init__(self):
        super().__init__()
        layers = [nn.Conv2d(3, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64), nn.ReLU(), nn.MaxPool2d(3, 2, 1)]
        c = 64
        for mid, n, stride in ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)):
            for u in range(n):
                layers.append(_Bottleneck(c, mid, stride if u == 0 else 1))
                c = mid * 4
        self.trunk = nn.Sequential(*layers)
        self.head = nn.Sequential(nn.Linear(2048, 1024), nn.ReLU(), nn.Linear(1024, self.N_PARAM))
        self.landmarks = nn.Linear(100 + 50 + 6, 68 * 2)        # fixed random "FLAME": (shape, exp, pose) -> 68 2-D landmarks

    def forward(self, images):
        """images [B,3,256,256] in [-1,1] -> (params dict, angles [B,3] in degrees) like calculate_shapemodel."""
        x = F.interpolate(images, size=(224, 224), mode='bilinear', align_corners=False)
        p = self.head(self.trunk(x).mean((2, 3)))
        shp, exp, pose, cam = p[:, :100], p[:, 150:200], p[:, 200:206], p[:, 206:209]
        angles = pose[:, :3] * 57.29577951308232                 # the reference converts the axis-angle head pose to Euler degrees
        return {'pose': pose, 'alpha_exp': exp, 'alpha_shp': shp, 'cam': cam}, angles

    def landmark_loss(self, gt, reen):
        """L1 between the stand-in landmarks of the ground-truth coefficient mix and of the reenacted image
        (utils_train.py:383-419 computes shape / mouth / eye L1 terms on FLAME landmarks)."""
        def lm(d):
            return self.landmarks(torch.cat([d['alpha_shp'], d['alpha_exp'], d['pose']], 1))
        return (lm(gt).detach() - lm(reen)).abs().mean()
