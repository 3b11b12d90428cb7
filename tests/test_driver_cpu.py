"""Host logic of the minimal drivers (CPU): PSNR formula and image assembly as Evaluator.evaluate."""
import numpy as np
import torch

from invr import driver


def test_psnr_metric_and_image_assembly():
    H, W = 4, 5
    mask = np.zeros((H, W), bool)
    mask[1:3, 1:4] = True
    batch = {'mask_at_box': torch.from_numpy(mask.reshape(1, -1)), 'H': torch.tensor([H]), 'W': torch.tensor([W])}
    vals = np.arange(6 * 3, dtype=np.float64).reshape(6, 3) / 20
    img = driver.assemble_image(vals, batch)
    assert img.shape == (H, W, 3) and img[0].sum() == 0 and np.allclose(img[mask], vals)
    gt = np.zeros_like(img)
    mse = np.mean((img - gt) ** 2)
    assert np.isclose(driver.psnr_metric(img.reshape(-1, 3), gt.reshape(-1, 3)), -10 * np.log10(mse))


def test_assemble_patch_matches_boolean_index_assignment():
    """inb_trainer.py:196-203 re-assembles the rays of a patch with `img[mask_at_box] = rgb`; the wrapper's sync-free gather
    must give the same image and pass the gradient to every ray."""
    from invr.trainer import assemble_patch
    g = torch.Generator().manual_seed(0)
    for H, W, p in ((7, 9, 0.6), (64, 64, 0.95), (5, 5, 0.0), (3, 4, 1.0)):
        mask = torch.rand(H * W, generator=g) < p
        n = int(mask.sum())
        vals = torch.rand(1, n, 3, generator=g, requires_grad=True)
        ref = torch.zeros(H, W, 3)
        ref[mask.reshape(H, W)] = vals.detach()[0]
        img = assemble_patch(vals, mask[None], H, W)
        assert torch.equal(img, ref)
        if n:
            img.sum().backward()
            assert vals.grad is not None and bool((vals.grad == 1).all())


def test_network_wrapper_refuses_silent_mse_for_lpips_configs():
    """configs/inb/inb_377.yaml sets use_lpips True: without a perceptual-loss module the wrapper must raise, never train MSE
    silently; with an injected module construction succeeds."""
    import pytest
    from invr.config import make_cfg
    from invr.network import Network
    from invr.trainer import NetworkWrapper
    from invr.losses import PerceptualLoss
    cfg = make_cfg(table_log2=8, use_lpips=True)
    net = Network(cfg=cfg)
    with pytest.raises(RuntimeError, match='use_lpips'):
        NetworkWrapper(net)
    w = NetworkWrapper(net, perceptual_loss=PerceptualLoss(allow_random=True))
    assert hasattr(w, 'perceptual_loss')
    with pytest.raises(RuntimeError, match='use_ssim'):
        NetworkWrapper(Network(cfg=make_cfg(table_log2=8, use_ssim=True)))


def test_perceptual_loss_structure_and_torchvision_keys():
    """perceptual_loss.py:6-68: relu1_2 / relu2_2 of VGG19 (features[3], features[8]); torchvision-keyed state dicts load."""
    import pytest
    from invr.losses import PerceptualLoss, VggRelu12
    with pytest.raises(RuntimeError):
        PerceptualLoss()
    src = VggRelu12()
    sd = {'features.' + k: v + 0.01 for k, v in src.vgg_layers.state_dict().items()}
    sd['features.10.weight'] = torch.zeros(1)                  # deeper layers of a full checkpoint are ignored
    sd['classifier.0.weight'] = torch.zeros(1)
    pl = PerceptualLoss(weights=sd)
    assert torch.equal(pl.model.vgg_layers[5].weight, src.vgg_layers[5].weight + 0.01)
    x, t = torch.rand(1, 3, 64, 64, requires_grad=True), torch.rand(1, 3, 64, 64)
    f = pl.model(x)
    assert f[0].shape == (1, 64, 64, 64) and f[1].shape == (1, 128, 32, 32)
    loss = pl(x, t)
    loss.backward()
    assert x.grad is not None and float(loss) > 0
    assert float(pl(t, t)) == 0.0
