"""`MetricsCalculator` -- evaluation metrics with the reference's interface (/root/reference/src/metrics.py:150-386).

Evaluation only (not part of the denoising hot path): plain torch ops, as the reference's
torchmetrics are.  Restated here: SSIM (torchmetrics defaults: 11x11 Gaussian, sigma 1.5, k1/k2 0.01/0.03,
data_range 1), PSNR and MSE, all on 512x512 LANCZOS-resized RGB in [0,1] (reference :227-239, :291-347).
LPIPS(squeeze), CLIPScore(ViT-B/16) and the DINO ViT-B/8 distance need checkpoints that only exist on the hub
(reference :28, :179-186); offline they return None rather than a made-up number."""
import numpy as np
import torch
import torch.nn.functional as F
from PIL import Image

TARGET = (512, 512)


class MetricsCalculator:
    def __init__(self, device="cuda"):
        self.device = device
        print(f"[MetricsCalculator] Initializing on {device}...")
        d = torch.arange(-5.0, 6.0)
        g = torch.exp(-(d / 1.5) ** 2 / 2)
        g = (g / g.sum())[None]
        self._win = (g.T @ g).to(device)
        print("[MetricsCalculator] Initialization complete! (LPIPS / CLIP score / DINO: unavailable offline)")

    def _pil_to_tensor(self, img):
        a = np.array(img).astype(np.float32) / 255.0
        return torch.from_numpy(a).permute(2, 0, 1).unsqueeze(0).to(self.device)

    def _pair(self, img1, img2):
        if img1.size != TARGET:
            img1 = img1.resize(TARGET, Image.LANCZOS)
        if img2.size != TARGET:
            img2 = img2.resize(TARGET, Image.LANCZOS)
        return self._pil_to_tensor(img1), self._pil_to_tensor(img2)

    def calculate_ssim(self, img1, img2):
        x, y = self._pair(img1, img2)
        c = x.shape[1]
        win = self._win.expand(c, 1, 11, 11)
        xp, yp = F.pad(x, (5,) * 4, mode="reflect"), F.pad(y, (5,) * 4, mode="reflect")
        mu_x, mu_y, e_xx, e_yy, e_xy = F.conv2d(torch.cat([xp, yp, xp * xp, yp * yp, xp * yp]), win, groups=c).split(1)
        c1, c2 = 0.01 ** 2, 0.03 ** 2
        sxx, syy, sxy = e_xx - mu_x ** 2, e_yy - mu_y ** 2, e_xy - mu_x * mu_y
        m = ((2 * mu_x * mu_y + c1) * (2 * sxy + c2)) / ((mu_x ** 2 + mu_y ** 2 + c1) * (sxx + syy + c2))
        return m[..., 5:-5, 5:-5].mean().item()

    def calculate_mse(self, img1, img2):
        x, y = self._pair(img1, img2)
        return ((x - y) ** 2).mean().item()

    def calculate_psnr(self, img1, img2):
        m = self.calculate_mse(img1, img2)
        return float("inf") if m == 0 else float(10.0 * np.log10(1.0 / m))

    def calculate_lpips(self, img1, img2):
        return None

    def calculate_clip_score(self, img, text):
        return None

    def calculate_all_metrics(self, source_img, edited_img, prompt):
        return {"ssim": self.calculate_ssim(source_img, edited_img), "lpips": self.calculate_lpips(source_img, edited_img),
                "clip_score": self.calculate_clip_score(edited_img, prompt), "psnr": self.calculate_psnr(source_img, edited_img),
                "mse": self.calculate_mse(source_img, edited_img), "dino_distance": None}

    def clear_memory(self):
        if self.device == "cuda":
            torch.cuda.empty_cache()
