"""Host logic of run_batch.py without a GPU: the per-image loop, --batch_size grouping, --in_flight worker threads, per-image
failure isolation (reference run_batch.py:176-261) -- driven through a stub editor."""
import os
import threading

import numpy as np
import pytest
from PIL import Image

import run_batch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class StubEditor:
    """Implements the slice of FastEditor that run_batch touches; records how it was called."""

    def __init__(self, fail_on=()):
        self.calls, self.batches, self.slots, self.fail_on = [], [], set(), set(fail_on)
        self.in_flight = 1
        self._tls = threading.local()
        self._lock = threading.Lock()

    def _out(self, prompt):
        if prompt in self.fail_on:
            raise RuntimeError(f"boom on {prompt}")
        return Image.fromarray(np.full((8, 8, 3), len(prompt) % 251, dtype=np.uint8))

    def edit(self, image, prompt, **kw):
        with self._lock:
            self.calls.append(prompt)
            self.slots.add(getattr(self._tls, "slot", 0))
        return self._out(prompt)

    def edit_batch(self, images, prompts, **kw):
        with self._lock:
            self.batches.append(list(prompts))
        return [self._out(p) for p in prompts]

    def set_in_flight(self, n):
        self.in_flight = n

    def worker_slot(self, slot):
        self._tls.slot = slot

    def clear_memory(self):
        pass


@pytest.fixture
def dataset(tmp_path):
    src = tmp_path / "src"
    entries = []
    for i in range(7):
        rel = f"{i % 3}_cat/{i:012d}.png"
        (src / f"{i % 3}_cat").mkdir(parents=True, exist_ok=True)
        Image.fromarray(np.full((16, 16, 3), i, dtype=np.uint8)).save(src / rel)
        entries.append((i, f"{i:012d}", {"image_path": rel, "editing_prompt": f"prompt {'x' * i}", "editing_type_id": str(i % 3)}))
    entries.append((7, "nosrc", {"image_path": "0_cat/missing.png", "editing_prompt": "y", "editing_type_id": "0"}))
    entries.append((8, "noprompt", {"image_path": "0_cat/000000000000.png", "editing_prompt": "", "editing_type_id": "0"}))
    entries.append((9, "evil", {"image_path": "../../etc/passwd", "editing_prompt": "z", "editing_type_id": "0"}))
    return src, entries


def _args(src, out, *extra):
    return run_batch.build_parser().parse_args(["--source_dir", str(src), "--output_dir", str(out), "--seed", "42"] + list(extra))


def test_serial_loop_counts_and_outputs(dataset, tmp_path):
    src, entries = dataset
    ed = StubEditor()
    r = run_batch.process_shard(ed, entries, _args(src, tmp_path / "o"), str(tmp_path / "o" / "e"), str(tmp_path / "o" / "c"))
    assert (r["processed"], r["skipped"], r["failed"]) == (7, 0, 3)
    assert [row["index"] for row in r["rows"]] == list(range(7)) and len(ed.calls) == 7 and not ed.batches
    assert all((tmp_path / "o" / "e" / e["image_path"]).exists() for _, _, e in entries[:7])
    r2 = run_batch.process_shard(ed, entries, _args(src, tmp_path / "o", "--skip_existing"), str(tmp_path / "o" / "e"), str(tmp_path / "o" / "c"))
    assert (r2["processed"], r2["skipped"], r2["failed"]) == (0, 8, 2)       # "noprompt" now finds image 0's output: skipped


def test_batched_loop_groups_and_flushes_the_tail(dataset, tmp_path):
    src, entries = dataset
    ed = StubEditor()
    r = run_batch.process_shard(ed, entries, _args(src, tmp_path / "o", "--batch_size", "3"), str(tmp_path / "o" / "e"), str(tmp_path / "o" / "c"))
    assert (r["processed"], r["failed"]) == (7, 3) and not ed.calls
    assert [len(b) for b in ed.batches] == [3, 3, 1]
    assert [row["index"] for row in r["rows"]] == list(range(7))


def test_failure_is_isolated_per_image_or_per_batch(dataset, tmp_path):
    src, entries = dataset
    bad = entries[4][2]["editing_prompt"]
    r = run_batch.process_shard(StubEditor(fail_on=[bad]), entries, _args(src, tmp_path / "a"), str(tmp_path / "a" / "e"), str(tmp_path / "a" / "c"))
    assert (r["processed"], r["failed"]) == (6, 4)
    r = run_batch.process_shard(StubEditor(fail_on=[bad]), entries, _args(src, tmp_path / "b", "--batch_size", "3"),
                                str(tmp_path / "b" / "e"), str(tmp_path / "b" / "c"))
    assert (r["processed"], r["failed"]) == (4, 6)                           # the failing batch (images 3-5) fails as a whole


def test_in_flight_workers_cover_every_entry_once(dataset, tmp_path):
    src, entries = dataset
    ed = StubEditor()
    r = run_batch.process_shard(ed, entries, _args(src, tmp_path / "o", "--in_flight", "2"), str(tmp_path / "o" / "e"), str(tmp_path / "o" / "c"))
    assert ed.in_flight == 2 and ed.slots == {0, 1}
    assert (r["processed"], r["failed"]) == (7, 3) and sorted(ed.calls) == sorted(e["editing_prompt"] for _, _, e in entries[:7])
    assert [row["index"] for row in r["rows"]] == list(range(7))


def test_bench_reads_the_pmc_record_of_the_tile_it_ran(tmp_path):
    """bench.py roofline.traffic (contract: PMC bytes of the dominant kernel, or null): the committed record holds one entry per tile code the
    autotuner may pick for the FF1 shape; the kernel name of the run selects it; an unknown tile, another shape, the fp8-weight kernels or
    a missing file give null instead of some other kernel's bytes."""
    import json

    import bench
    name = "gemm3_kernel<192x128> (gemm, tile code 54) (FF1 GEGLU projection, 32x32 latents)"
    mb, src = bench.pmc_traffic(name, 2048, 10240, 1280)
    assert mb is not None and 50 < mb < 400 and src and all(s.startswith("profiles/r03_ff1_pmc_raw/") and os.path.exists(os.path.join(ROOT, s)) for s in src)
    mb63, _ = bench.pmc_traffic(name.replace("<192x128>", "<256x320>").replace("code 54", "code 63"), 2048, 10240, 1280)    # round 3's exact-fit tile
    assert mb63 is not None and 50 < mb63 < mb
    assert bench.pmc_traffic(name.replace("code 54", "code 43"), 2048, 10240, 1280) == (None, None)
    assert bench.pmc_traffic(name, 4096, 10240, 1280) == (None, None)
    assert bench.pmc_traffic("gemm3w8_kernel<128x128> (gemm, fp8 weights, tile code 54)", 2048, 10240, 1280) == (None, None)
    assert bench.pmc_traffic(name, 2048, 10240, 1280, path=str(tmp_path / "absent.json")) == (None, None)
    single = tmp_path / "single.json"                        # the single-record form of round 1
    single.write_text(json.dumps({"shape": {"M": 8, "N": 8, "K": 8}, "fetch_size_kib": 1000.0, "write_size_kib": 48.0, "source": ["x.csv"]}))
    assert bench.pmc_traffic("any kernel (tile code 1)", 8, 8, 8, path=str(single)) == (round((2 * 1000.0 + 48.0) * 1024 / 1e6, 1), ["x.csv"])


@pytest.mark.parametrize("geglu", [False, True])
def test_layernorm_fold_tables_reproduce_layernorm_plus_linear(geglu):
    """The algebra behind fie_gemm_ln_f16 (hip.fold_layernorm_tables), on the CPU in float64: rstd * (x Wf^T - mean * S) + b' against LayerNorm -> Linear
    with the f16-rounded folded weights, rows with a common offset several sigma large; GEGLU: the table follows the packed (value, gate interleaved) row order."""
    import torch
    import torch.nn.functional as F
    from fie_amd.hip import fold_layernorm_tables
    g = torch.Generator().manual_seed(3)
    m, n, k = 37, 64, 128
    x = (torch.randn(m, k, generator=g) * 2 + torch.randn(m, 1, generator=g) * 6).half()
    w, b = (torch.randn(n, k, generator=g) / k ** 0.5).half(), (torch.randn(n, generator=g) * 0.1).half()
    gamma, beta = (1 + 0.2 * torch.randn(k, generator=g)).half(), (0.1 * torch.randn(k, generator=g)).half()
    wf, tab = fold_layernorm_tables(w, b, gamma, beta, geglu=geglu)
    assert wf.dtype == torch.float16 and tab.dtype == torch.float32 and tab.shape == (n, 2)
    xd = x.double()
    mean, var = xd.mean(1, keepdim=True), xd.var(1, unbiased=False, keepdim=True)
    rstd = (var + 1e-5).rsqrt()
    order = torch.stack([torch.arange(n // 2), torch.arange(n // 2) + n // 2], 1).reshape(-1) if geglu else torch.arange(n)      # packed row -> source row
    folded = rstd * (xd @ wf.double()[order].t() - mean * tab[:, 0].double()[None, :]) + tab[:, 1].double()[None, :]
    # the same LayerNorm -> Linear with the weights the kernel multiplies by (W * gamma rounded to f16; beta and bias exact)
    ref = ((xd - mean) * rstd) @ wf.double()[order].t() + (w.double() @ beta.double() + b.double())[order][None, :]
    assert (folded - ref).abs().max() < 1e-5                                          # the table is fp32
    plain = F.layer_norm(xd, (k,), gamma.double(), beta.double(), 1e-5) @ w.double().t() + b.double()
    assert (folded - plain[:, order]).abs().max() / plain.abs().max() < 2e-3          # what rounding W * gamma to f16 costs
