"""Image-space loss of the reference's LPIPS branch (lib/train/trainers/loss/perceptual_loss.py:6-68): L1 on the relu1_2 /
relu2_2 activations of a frozen VGG19 + L1 + L2 on the image.

The reference builds the VGG through torchvision and downloads ImageNet weights; neither torchvision nor a network is
available on this image, so the feature stack is restated here as plain `nn.Conv2d`s with torchvision's own
`vgg19().features` indices and state_dict keys ('0.weight', '2.weight', '5.weight', '7.weight'): a torchvision
checkpoint (full model or its `.features`) loads directly.  Without weights the constructor refuses to build a loss
network (training against random features would silently optimise a different objective) unless `allow_random=True`
(tests of the patch re-assembly / plumbing only).
"""
import torch
import torch.nn as nn


class VggRelu12(nn.Module):
    """features[0..8] of torchvision's vgg19: conv3-64, relu, conv64-64, relu (3: "relu1"), maxpool, conv64-128, relu,
    conv128-128, relu (8: "relu2") — perceptual_loss.py:36-47 stops after layer 8."""

    def __init__(self):
        super().__init__()
        self.vgg_layers = nn.Sequential(
            nn.Conv2d(3, 64, 3, padding=1), nn.ReLU(inplace=False), nn.Conv2d(64, 64, 3, padding=1), nn.ReLU(inplace=False),
            nn.MaxPool2d(2, 2), nn.Conv2d(64, 128, 3, padding=1), nn.ReLU(inplace=False), nn.Conv2d(128, 128, 3, padding=1),
            nn.ReLU(inplace=False))
        for p in self.parameters():
            p.requires_grad = False

    def load_torchvision(self, sd):
        """Accepts vgg19().state_dict() ('features.N.*'), vgg19().features.state_dict() ('N.*') or this module's own."""
        own = {}
        for k, v in sd.items():
            k = k[len('features.'):] if k.startswith('features.') else k
            k = k[len('vgg_layers.'):] if k.startswith('vgg_layers.') else k
            if k.split('.')[0] in ('0', '2', '5', '7'):
                own[k] = v
        self.vgg_layers.load_state_dict(own, strict=True)
        return self

    def forward(self, x):
        out = []
        for i, m in enumerate(self.vgg_layers):
            x = m(x)
            if i in (3, 8):
                out.append(x)
        return out


class PerceptualLoss(nn.Module):
    """perceptual_loss.py:45-68: (L1(relu1) + L1(relu2)) / 2 + L1(image) + MSE(image); inputs (1,3,H,W)."""

    def __init__(self, weights=None, allow_random=False):
        super().__init__()
        self.model = VggRelu12()
        if weights is not None:
            sd = torch.load(weights, map_location='cpu') if isinstance(weights, str) else weights
            self.model.load_torchvision(sd)
        elif not allow_random:
            raise RuntimeError('PerceptualLoss needs the VGG19 ImageNet weights (torchvision vgg19 state_dict); pass weights=<path or '
                               'state_dict>.  There is no network on this image to download them.')
        self.model.eval()

    def forward(self, x, target):
        fx, ft = self.model(x[:, 0:3]), self.model(target[:, 0:3])
        feature_loss = ((fx[0] - ft[0]).abs().mean() + (fx[1] - ft[1]).abs().mean()) / 2.0
        return feature_loss + (x - target).abs().mean() + ((x - target) ** 2).mean()
