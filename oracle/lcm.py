"""ORACLE -- TEST INFRASTRUCTURE ONLY.

LCMScheduler restated (upstream diffusers 0.35.2 schedulers/scheduling_lcm.py; installed by the reference at
/root/reference/src/pipeline.py:138-141,158-161).  Closed forms follow SURVEY.md A.5; pinned by the known
answers listed there (timesteps [999,759,499,259]; alpha-bar table) in tests/test_oracle_cpu.py.
"""
import numpy as np
import torch


class LCMOracle:
    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012,
                 original_inference_steps=50, timestep_scaling=10.0, sigma_data=0.5, **_):
        self.T = num_train_timesteps
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = self.alphas_cumprod[0]      # set_alpha_to_one=False
        self.original_steps = original_inference_steps
        self.timestep_scaling = timestep_scaling
        self.sigma_data = sigma_data
        self.timesteps = None
        self.num_inference_steps = None
        self.step_index = None

    def set_timesteps(self, n, strength=1.0):
        """scheduling_lcm.py::set_timesteps.  The SDXL ControlNet img2img pipeline calls it WITHOUT strength
        (SURVEY 0 item 2), i.e. strength=1.0 here, and trims afterwards with get_timesteps()."""
        k = self.T // self.original_steps
        origin = np.asarray(list(range(1, int(self.original_steps * strength) + 1))) * k - 1
        origin = origin[::-1].copy()
        idx = np.floor(np.linspace(0, len(origin), num=n, endpoint=False)).astype(np.int64)
        self.timesteps = [int(v) for v in origin[idx]]
        self.num_inference_steps = n
        self.step_index = None
        return self.timesteps

    def get_timesteps(self, n, strength):
        """pipeline_controlnet_sd_xl_img2img.py::get_timesteps: drop the first n - int(n*strength) steps."""
        init = min(int(n * strength), n)
        t_start = max(n - init, 0)
        self.begin_index = t_start
        self.step_index = t_start
        return self.timesteps[t_start:], n - t_start

    def add_noise(self, x0, noise, t):
        a = self.alphas_cumprod[t]
        return a.sqrt() * x0 + (1.0 - a).sqrt() * noise

    def boundary(self, t):
        s = t * self.timestep_scaling
        c_skip = self.sigma_data ** 2 / (s ** 2 + self.sigma_data ** 2)
        c_out = s / (s ** 2 + self.sigma_data ** 2) ** 0.5
        return c_skip, c_out

    def step(self, eps, t, x, noise=None):
        """scheduling_lcm.py::step, prediction_type='epsilon', no clipping/thresholding."""
        i = self.step_index
        prev_t = self.timesteps[i + 1] if i + 1 < len(self.timesteps) else t
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        c_skip, c_out = self.boundary(t)
        x0 = (x - (1.0 - a_t).sqrt() * eps) / a_t.sqrt()
        den = c_out * x0 + c_skip * x
        if i != self.num_inference_steps - 1:
            assert noise is not None
            out = a_prev.sqrt() * den + (1.0 - a_prev).sqrt() * noise
        else:
            out = den
        self.step_index = i + 1
        return out, den
