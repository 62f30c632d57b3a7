"""Driver-facing contracts: bench.py's JSON line, __graft_entry__.smoke(), launch enumeration, size limits."""
import json
import os
import subprocess
import sys

import pytest
import torch

from conftest import ROOT
from emavfi import EMA_VFI, lib, synth


def test_forward_launch_enumeration_matches_survey_flops():
    """1 054 908 FLOP/px in the reference's formulation; this build folds the
    broadcast-context half of motion_estimation.0 (2*36 864 FLOP/px) into a bias and adds ~24 FLOP/px of warp."""
    B, H, W = 2, 64, 96
    launches = lib.forward_launches(3, 64, 3, B, H, W, "bf16")
    names = [n for n, _, _ in launches]
    # bf16 at the reference width: each ModulatedDeformConvPack (offset_conv + dcn_v2) is ONE launch (23 - 3),
    # cat(frame1, frame2) + feat_ext_conv1 + conv_block_0 is one launch (no pack_input, no conv_block_0: - 2) and so is
    # motion_estimation.1 + .2 (- 1), reconstruction.1 + .2 (- 1) and, since round 4, conv_block_1 + conv_block_2 (- 1); + the
    # blob-header guard as the last launch (round 4)
    assert len(launches) == 16 and sum(n.startswith("conv3x3+conv3x3<bf16,64->64->64>") for n in names) == 1 and sum("+head" in n for n in names) == 1 and names[0].startswith("conv_first+conv3x3<bf16,6->64->64>") and names[-2].startswith("conv3x3+tail<bf16,64->32->3>") and names[-1] == "blob_guard"
    assert sum(n.startswith("deform<") and n.endswith("offset_conv+dcn_v2") for n in names) == 3
    per_px = sum(f for _, f, _ in launches) / (B * H * W)
    assert abs(per_px - (1054908 - 2 * 36864 + 24)) < 1.0
    assert all(b > 0 for _, _, b in launches)
    # fp32 keeps the two launches per pack; labels and byte counts differ, the flop total does not
    l32 = lib.forward_launches(3, 64, 3, B, H, W, "fp32")
    assert len(l32) == 24 and sum(n.startswith("deform<") for n, _, _ in l32) == 3
    assert abs(sum(f for _, f, _ in l32) - sum(f for _, f, _ in launches)) < 1.0
    assert all("f32" in n for n, _, _ in l32 if "<" in n)


def test_size_limits_are_argument_errors_not_crashes():
    L = lib.load()
    rc = L.emavfi_forward(3, 64, 3, None, 0, None, None, None, None, 0, 1, 4096, 4096, lib.BF16, None, None)
    assert rc == -1 and "2^24" in lib.last_error()
    rc = L.emavfi_forward(3, 64, 3, None, 0, None, None, None, None, 0, 1, 8192, 4096, lib.F32, None, None)
    assert rc == -1 and "4 GiB" in lib.last_error()
    # EMAVFI_AMP16 runs the fp32 deformable kernel on an fp32 fusion tensor (320-byte pixels) although its conv kernels are 16-bit:
    # 5120x2880 = 14.7M pixels passes the 2^24 pixel guard and the 2-byte plane guard but wraps 32-bit byte offsets at 4 bytes
    rc = L.emavfi_forward(3, 64, 3, None, 0, None, None, None, None, 0, 1, 2880, 5120, lib.AMP16, None, None)
    assert rc == -1 and "4 GiB" in lib.last_error()
    rc = L.emavfi_forward(3, 64, 3, None, 0, None, None, None, None, 0, 1, 2880, 5120, lib.F16, None, None)
    assert rc == -1 and "null pointer" in lib.last_error()     # the same size is fine for the all-16-bit mode (fails later, on the nulls)
    # stage-level conv entry: 32-bit DMA source offsets inside one sample (refused before anything is launched or dereferenced)
    import ctypes
    fake = ctypes.c_void_p(256)
    rc = L.emavfi_conv3x3(fake, fake, fake, fake, 1, 64, 64, 32768, 32768, 1, 0, lib.BF16, fake, 0, None)
    assert rc == -1 and "4 GiB" in lib.last_error()
    assert L.emavfi_forward_launches(3, 64, 3, 1, 64, 64, lib.BF16, None, 0, None, None, 0) == 16
    assert L.emavfi_forward_launches(3, 64, 3, 1, 64, 64, lib.F32, None, 0, None, None, 0) == 24
    assert L.emavfi_forward_launches(3, 7, 3, 1, 64, 64, lib.BF16, None, 0, None, None, 0) == -2


@pytest.mark.gpu
def test_bench_json_contract():
    """`python bench.py` prints ONE JSON line with the fields the driver and the judge read."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "2",
                          "--height", "96", "--width", "128", "--cpu-reps", "2"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    r = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in r, k
    assert r["n_gpus"] == 1 and r["steps"] == 2 and r["warmup"] == 1 and r["higher_is_better"] is True
    assert r["scaling"] == "weak" and r["vs_baseline"] is None and r["dtype"] == "bf16" and r["data"] == "synthetic"
    assert "workload" in r["config"] and "model" not in r["config"]
    assert r["value"] > 0 and abs(r["value"] - 2 * 2 / (r["ms_per_step"] * 2e-3)) / r["value"] < 0.02
    rf = r["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] in ("GB/s", "TFLOP/s") and "traffic" in rf
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert rf["traffic"] is None or rf["traffic"] > 0
    assert "traffic_source" in rf     # PMC bytes are refused when profiles/traffic.json was measured on other kernel sources
    # round 6: where the headline sits on the pack's offset curve - the kernels' own counters of the last timed forward
    assert len(rf["fixup_share"]) == 3 and all(0.0 <= v <= 1.0 for v in rf["fixup_share"])
    assert all(pp["wave_taps"] == 2 * 6 * 8 * 4 * 9 for pp in rf["offset_census"]["per_pack"])      # B = 2, 96 x 128: 6 x 8 tiles x 4 waves x 9 taps
    cb = r["cpu_baseline"]
    assert 0.3 <= cb["dcn_restatement_share"] <= 0.99 and cb["gflops_aten_only"] > cb["gflops"] and cb["cores"] <= 16
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb and cb["unit"] == "frames/s"
    assert cb["cpu_model"] and cb["also_256x256"]["value"] > 0 and "1 full-size warm-up + 2 timed" in cb["sample"]
    assert cb["also_config1_b16_256x256"]["value"] > 0
    assert r["accuracy_vs_cpu_oracle"]["fp32"]["max_abs"] <= 1e-3
    assert r["accuracy_vs_cpu_oracle"]["fp16"]["psnr_db"] >= r["accuracy_vs_cpu_oracle"]["bf16"]["psnr_db"]
    # reported beside `value`, never instead of it: the other arithmetic modes at the same size, BASELINE configs[1]
    # (batch 16 of 256x256) in fp32 and bf16, and the warp kernel the forward itself runs
    assert r["accuracy_vs_cpu_oracle"]["fp32x3"]["max_abs"] <= 1e-3      # the fp32-accurate split mode under the fp32 gate
    for k in ("also_fp16_fast", "also_amp16_autocast_policy", "also_fp32_exact", "also_fp32_split16"):
        assert r[k]["value"] > 0 and r[k]["height"] == 96, k
    assert r["config1_256"]["fp32"]["value"] > 0 and r["config1_256"]["bf16"]["pairs_per_step"] == 16
    # the exact-fp32 mode (the one north_star's 1e-3 bound is about) with ITS dominant kernel's roofline, at both sizes
    for k in ("roofline_fp32", "roofline_fp32_config1"):
        rr = r[k]
        assert rr["bound"] in ("hbm", "mfma") and rr["peak"] in (157.3, 8000.0) and abs(rr["frac"] - rr["achieved"] / rr["peak"]) < 1e-3 and "workload" in rr
    assert r["ranks_seen"] == [0] and len(r["ms_per_step_per_rank"]) == 1
    wf = r["roofline_warp_in_forward"]
    assert wf["layout_bytes_per_launch"] > wf["algorithmic_bytes_per_launch"] and wf["frac_on_layout_bytes"] > wf["frac"]
    assert r["roofline_warp_in_forward"]["bound"] == "hbm" and r["roofline_warp_in_forward"]["kernel"].startswith("warp_fused")
    # package power / clocks while the forward runs back to back ({} if rocm-smi is not usable on the box)
    bl = r["board_under_load"]
    assert isinstance(bl, dict) and (not bl or (bl["samples"] >= 1 and bl.get("sclk_mhz", 1) > 0))


@pytest.mark.gpu
def test_bench_two_ranks_the_way_the_driver_calls_it():
    """`python bench.py --gpus 2 ...` (no torchrun environment): the parent launches the ranks itself.  Two ranks share this box's
    one GPU over gloo (RCCL refuses two ranks on one device; EMAVFI_DIST_BACKEND selects the backend, the driver's node uses the
    default "nccl"): rank 1 runs from the blob rank 0 broadcast (verified by emavfi_packed_check on arrival)."""
    env = dict(os.environ, EMAVFI_DIST_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "1",
                          "--height", "96", "--width", "128"], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["ranks_seen"] == [0, 1] and len(r["ms_per_step_per_rank"]) == 2 and r["backend"] == "gloo"
    assert r["value"] > 0 and abs(r["value"] - 2 * 1 * 2 / (r["ms_per_step"] * 2e-3)) / r["value"] < 0.02   # whole-job frames / MAX time
    assert "roofline" in r and "cpu_baseline" not in r        # the CPU baseline is rank 0's at N = 1 only
    # round 6: the PCIe-inclusive leg per rank - every rank streams ITS 64-pair segment of one 129-frame stream
    sp = r["also_stream_pcie_per_rank"]
    assert sp["pairs_per_rank"] == [64, 64] and sp["frames_out_per_rank"] == [128, 129] and len(sp["stream_pcie_per_rank"]) == 2
    assert all(v > 0 for v in sp["stream_pcie_per_rank"]) and sp["stream_pcie_aggregate"] > 0
    assert abs(sp["stream_pcie_aggregate"] - 128 / max(sp["seconds_per_rank"])) <= 0.02 * sp["stream_pcie_aggregate"]
    frac = sp["stream_pcie_aggregate"] / r["value"]      # (both printed rounded to two decimals: a relative tolerance)
    assert abs(sp["fraction_of_resident_value"] - frac) <= 2e-3 * max(1.0, frac)


@pytest.mark.gpu
def test_smoke_entry_point():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.smoke()


@pytest.mark.gpu
def test_config5_1080p_runs_deterministically():
    """BASELINE.json configs[4] size (1920x1080): ragged 16-row deform tiles (1080 = 67.5 x 16), bf16 vs fp32."""
    sd = synth.synthetic_state_dict(seed=0)
    f1, f2 = synth.fast_frames(9, 1, 1080, 1920, device="cuda:0")
    outs = {}
    for mode in ("fp32", "bf16"):
        m = EMA_VFI(compute_dtype=mode).cuda().eval()
        m.load_state_dict(sd)
        with torch.no_grad():
            a, b = m(f1, f2), m(f1, f2)
        assert torch.equal(a, b) and torch.isfinite(a).all() and a.min() >= 0 and a.max() <= 1
        outs[mode] = a
    mse = (outs["fp32"].double() - outs["bf16"].double()).pow(2).mean().item()
    assert mse < 10 ** (-5.0)  # PSNR of bf16 against the exact-fp32 frame > 50 dB (measured ~59 dB)


def test_amp16_launch_enumeration_has_no_rounding_passes():
    """Round 5: in the autocast-policy mode the fp32 DCN's epilogue and the warp write the fp16 roundings themselves -
    the per-block `fusion_round` passes and `fusion_round_warped` are gone at the reference width (the widening of `feat` stays)."""
    names = [n for n, _, _ in lib.forward_launches(3, 64, 3, 2, 64, 96, "amp16")]
    assert not any(n.startswith("fusion_round") for n in names), names
    assert sum(n.startswith("fusion_widen_feat") for n in names) == 1
    assert sum("writes fp32 + its fp16 rounding" in n for n in names) == 3
    # a width the fp32 LDS-window kernel does not serve keeps the separate pass
    small = [n for n, _, _ in lib.forward_launches(3, 8, 3, 2, 64, 96, "amp16")]
    assert sum(n.startswith("fusion_round") for n in small) == 3


def test_fixup_census_restates_the_pack_kernels_window_test():
    """bench.py's also_pack_vs_offset_spread reports which deformable samples leave deform_pack3_kernel's staged window; its vectorised
    census is checked here against a per-sample loop that follows csrc/deform_pack3.inl line by line (tile origin - 3, clamp to
    [-2, size + 1], floor, corner row / column in [0, 21]; a wave = 4 rows x 16 columns of a 16 x 16 tile)."""
    import math
    import bench
    g = torch.Generator().manual_seed(3)
    H, W = 37, 50
    raw = torch.randn(1, 27, H, W, generator=g) * 2.5
    got = bench.fixup_census(raw, H, W)
    off = torch.cat([raw[:, 0:9], raw[:, 18:27]], 1)[0]
    out = 0
    groups = set()
    for y in range(H):
        for x in range(W):
            ty0, tx0 = y // 16 * 16 - 3, x // 16 * 16 - 3
            for k in range(9):
                i, j = divmod(k, 3)
                py = min(max(float(y - 1 + i) + off[2 * k, y, x].item(), -2.0), H + 1.0)
                px = min(max(float(x - 1 + j) + off[2 * k + 1, y, x].item(), -2.0), W + 1.0)
                ly, lx = math.floor(py) - ty0, math.floor(px) - tx0
                if not (0 <= ly <= 21 and 0 <= lx <= 21):
                    out += 1
                    groups.add((k, y // 4, x // 16))
    Hp, Wp = (H + 15) // 16 * 16, (W + 15) // 16 * 16
    assert abs(got["samples_outside_window"] - out / (9 * H * W)) < 1e-5
    assert abs(got["wave_taps_in_fixup_loop"] - len(groups) / (9 * (Hp // 4) * (Wp // 16))) < 1e-5
    assert out > 0 and len(groups) > 0


def test_bench_host_helpers(tmp_path, monkeypatch):
    """bench.py's host-side helpers of round 6 (no GPU): the CPU thread count of the oracle baseline = min(affinity, cgroup quota, share),
    and the summary of the per-rank PCIe leg (aggregate = all pairs / the SLOWEST rank's time)."""
    sys.path.insert(0, ROOT)
    import bench
    q = tmp_path / "cpu.max"
    q.write_text("400000 100000\n")
    monkeypatch.delenv("EMAVFI_CPU_THREADS", raising=False)
    n_aff = len(os.sched_getaffinity(0))
    assert bench.granted_cpu_threads(str(q)) == min(4, n_aff)                     # a 4-CPU quota caps a wider mask
    q.write_text("max 100000\n")
    assert bench.granted_cpu_threads(str(q)) == min(16, n_aff)                    # no quota: the pool's stated share
    monkeypatch.setenv("EMAVFI_CPU_THREADS", "2")
    assert bench.granted_cpu_threads(str(tmp_path / "missing")) == min(2, n_aff)
    s = bench.stream_leg_summary([[1.0, 64.0, 129.0, 0.5], [0.0, 64.0, 128.0, 0.4]], 1000.0)
    assert s["stream_pcie_per_rank"] == [160.0, 128.0] and s["stream_pcie_aggregate"] == 256.0 and s["fraction_of_resident_value"] == 0.256
    assert s["pairs_per_rank"] == [64, 64] and s["frames_out_per_rank"] == [128, 129]
