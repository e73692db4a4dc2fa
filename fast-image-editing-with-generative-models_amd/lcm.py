"""LCMScheduler host logic (upstream schedulers/scheduling_lcm.py, installed by the
reference at src/pipeline.py:138-141,158-161).  Only tables and scalars live here -- the per-element update runs
in the fie_latent_prep / fie_lcm_step HIP kernels.  `timestep_spacing="trailing"` is accepted and ignored, as
upstream's LCM set_timesteps does."""
import numpy as np
import torch


class LCMSchedule:
    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                 original_inference_steps=50, timestep_scaling=10.0, sigma_data=0.5, set_alpha_to_one=False,
                 timestep_spacing="leading", **_):
        if beta_schedule != "scaled_linear":
            raise ValueError(f"beta_schedule {beta_schedule!r} not supported")
        self.T = num_train_timesteps
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0).double().numpy()
        self.final_alpha_cumprod = 1.0 if set_alpha_to_one else float(self.alphas_cumprod[0])
        self.original_steps = original_inference_steps
        self.timestep_scaling = timestep_scaling
        self.sigma_data = sigma_data
        self.timestep_spacing = timestep_spacing

    def timesteps(self, n):
        if n > self.T or n > self.original_steps:
            raise ValueError(f"num_inference_steps={n} exceeds the schedule ({self.original_steps} original steps)")
        k = self.T // self.original_steps
        origin = (np.arange(1, self.original_steps + 1) * k - 1)[::-1]
        idx = np.floor(np.linspace(0, len(origin), num=n, endpoint=False)).astype(np.int64)
        return [int(v) for v in origin[idx]]

    def plan(self, n, strength):
        """Returns the step records the pipeline executes: get_timesteps() drops the first n - int(n*strength)
        entries; the LAST scheduler index (n-1) is the only step without fresh noise."""
        ts = self.timesteps(n)
        init = min(int(n * strength), n)
        t_start = max(n - init, 0)
        steps = []
        for i in range(t_start, n):
            t = ts[i]
            prev_t = ts[i + 1] if i + 1 < n else t
            a_t = self.alphas_cumprod[t]
            a_p = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
            s = t * self.timestep_scaling
            steps.append(dict(
                t=t, last=(i == n - 1),
                sqrt_ab=float(np.sqrt(a_t)), sqrt_1mab=float(np.sqrt(1 - a_t)),
                c_skip=float(self.sigma_data ** 2 / (s ** 2 + self.sigma_data ** 2)),
                c_out=float(s / np.sqrt(s ** 2 + self.sigma_data ** 2)),
                sqrt_ab_prev=float(np.sqrt(a_p)), sqrt_1mab_prev=float(np.sqrt(1 - a_p))))
        return steps
