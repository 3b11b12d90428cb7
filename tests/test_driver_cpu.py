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
