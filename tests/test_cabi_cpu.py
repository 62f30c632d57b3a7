"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol
include/emavfi.h declares, answers its host-only queries, and the Python mirror keeps the
reference's state_dict contract.  No kernel is launched here."""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT
from emavfi import EMA_VFI, ModulatedDeformConvPack, lib, synth

RING2 = 1       # launches the two-layer ring fusions remove from the default bf16 plan (conv_block_1 + conv_block_2)


def header_symbols():
    text = open(os.path.join(ROOT, "include", "emavfi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(emavfi_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = lib.load()
    declared = header_symbols()
    assert len(declared) >= 12
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/emavfi.h but not exported"
    assert sorted(lib.SYMBOLS) == declared, "emavfi/lib.py prototypes out of sync with the header"
    assert L.emavfi_version() == 403


def test_host_queries_and_error_codes():
    L = lib.load()
    assert L.emavfi_param_count(3) == 40
    for mid in (8, 16, 32, 64):
        for dt in (lib.F32, lib.BF16, lib.F16):
            assert L.emavfi_supported(3, mid, 3, dt) == 0
            assert L.emavfi_packed_bytes(3, mid, 3, dt) > 0
    # round 6: the fp32-accurate split mode - its own dtype code, a plan on the generic tile kernels (three weight copies per chunk),
    # activations as two f16 halves (the workspace of the autocast-policy mode and more), launches enumerable without a GPU
    for mid in (8, 64):
        assert L.emavfi_supported(3, mid, 3, lib.F32X3) == 0
    assert lib.dtype_code("fp32x3") == lib.F32X3 == 4 and "#define EMAVFI_F32X3 4\n" in open(os.path.join(ROOT, "include", "emavfi.h")).read()
    assert L.emavfi_packed_bytes(3, 64, 3, lib.F32X3) > L.emavfi_packed_bytes(3, 64, 3, lib.AMP16)
    assert L.emavfi_workspace_bytes(3, 64, 3, 2, 96, 128, lib.F32X3) > L.emavfi_workspace_bytes(3, 64, 3, 2, 96, 128, lib.AMP16)
    names = [n for n, _, _ in lib.forward_launches(3, 64, 3, 2, 96, 128, "fp32x3")]
    assert sum("deform<f32" in n for n in names) == 3 and "fusion_split_warped" in names and not any("ring" in n or "+" in n.split(" ")[0] for n in names)
    assert L.emavfi_mdcn_workspace_bytes(1, 67, 8, 8, lib.F32X3, 0) == 0 and "F32X3" in lib.last_error()
    assert L.emavfi_supported(3, 7, 3, lib.F32) == -2 and "multiple of 8" in lib.last_error()
    assert L.emavfi_supported(3, 64, 3, 5) == -2
    assert L.emavfi_supported(3, 24, 3, lib.F32) == -2  # 27 -> 32 wide fusion has kernels, 96-wide context does not
    assert L.emavfi_packed_bytes(3, 64, 0, lib.F32) == 0
    # fp32 blob >= raw parameter bytes (padding), bf16 about half
    raw = sum(int(torch.tensor(s).prod()) for s in synth.param_shapes().values()) * 4
    assert L.emavfi_packed_bytes(3, 64, 3, lib.F32) >= raw
    assert L.emavfi_packed_bytes(3, 64, 3, lib.BF16) < L.emavfi_packed_bytes(3, 64, 3, lib.F32)
    # the two 16-bit modes share every size; the dtype codes are the header's
    # (at the reference width both blobs carry the three offset_conv layers a second time, in the one-launch pack kernel's
    # own fragment layout - csrc/deform_pack3.inl; the bf16 blob holds the bf16-rounded values as f16 there)
    assert L.emavfi_packed_bytes(3, 64, 3, lib.BF16) == L.emavfi_packed_bytes(3, 64, 3, lib.F16)
    assert L.emavfi_packed_bytes(3, 8, 3, lib.BF16) == L.emavfi_packed_bytes(3, 8, 3, lib.F16)
    assert L.emavfi_workspace_bytes(3, 64, 3, 2, 96, 128, lib.F16) == L.emavfi_workspace_bytes(3, 64, 3, 2, 96, 128, lib.BF16)
    hdr = open(os.path.join(ROOT, "include", "emavfi.h")).read()
    for name, code in (("EMAVFI_F32", lib.F32), ("EMAVFI_BF16", lib.BF16), ("EMAVFI_F16", lib.F16)):
        assert f"#define {name} {code}\n" in hdr
    assert lib.dtype_code("fp16") == lib.F16 and lib.dtype_code("half") == lib.F16 and lib.dtype_code("bfloat16") == lib.BF16
    assert L.emavfi_workspace_bytes(3, 64, 3, 8, 720, 1280, lib.BF16) < 8 << 30
    assert L.emavfi_workspace_bytes(3, 64, 3, 0, 720, 1280, lib.BF16) == 0
    # argument validation happens before any device work
    assert L.emavfi_warp(None, None, None, 1, 3, 8, 8, None) == -1
    assert L.emavfi_forward(3, 64, 3, None, 0, None, None, None, None, 0, 1, 8, 8, lib.F32, None, None) == -1
    assert L.emavfi_conv3x3(None, None, None, None, 1, 3, 3, 8, 8, 1, 0, lib.F32, None, 0, None) == -1
    assert L.emavfi_conv3x3_workspace_bytes(1, 64, 64, 32, 32, 3, lib.F32) == 0


@pytest.mark.parametrize("mid", [8, 64])
def test_state_dict_contract(mid):
    """key names, order and shapes equal the reference's."""
    m = EMA_VFI(mid_channels=mid)
    shapes = synth.param_shapes(mid_channels=mid)
    assert [k for k, _ in m.named_parameters()] == list(shapes)
    assert list(m.state_dict()) == list(shapes)  # no extra buffers
    for k, p in m.named_parameters():
        assert tuple(p.shape) == shapes[k], k
    m.load_state_dict(synth.synthetic_state_dict(seed=3, mid_channels=mid), strict=True)
    assert lib.load().emavfi_param_count(m.num_blocks) == len(shapes)
    # reference init quirk kept: offset convs start at zero (ema_vfi.py:42-43)
    fresh = ModulatedDeformConvPack(mid + 3, mid + 3)
    assert fresh.offset_conv.weight.abs().max() == 0 and fresh.offset_conv.bias.abs().max() == 0
    assert fresh.out_channels == mid + 3


def test_reference_import_path_resolves_to_native_class():
    from src.models.ema_vfi import EMA_VFI as Shim
    assert Shim is EMA_VFI


def test_cpu_tensors_fail_loudly_no_fallback():
    m = EMA_VFI().eval()
    x = torch.zeros(1, 3, 16, 16)
    with torch.no_grad():
        with pytest.raises(RuntimeError, match="no CPU path"):
            m(x, x)
        with pytest.raises(RuntimeError, match="no CPU path"):
            m.warp(x, x, torch.zeros(1, 2, 16, 16))
    with pytest.raises(ValueError):
        with torch.no_grad():
            m(torch.zeros(1, 3, 16, 16), torch.zeros(1, 3, 16, 17))


def test_product_path_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "video-frame-interpolation_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".inl", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", text, flags=re.M), f
                assert "liboracle" not in text, f


def test_checkpoint_variants_load():
    """DataParallel prefix, fp16 tensors, wrapper dict, shape errors."""
    sd = synth.synthetic_state_dict(seed=5, mid_channels=8)
    ref = EMA_VFI(mid_channels=8)
    ref.load_state_dict(sd, strict=True)
    variants = {
        "module-prefix": {"module." + k: v for k, v in sd.items()},
        "wrapped": {"state_dict": dict(sd), "epoch": 3},
        "half": {k: v.half() for k, v in sd.items()},
    }
    for name, v in variants.items():
        m = EMA_VFI(mid_channels=8)
        m.load_state_dict(v, strict=True)
        for (k, a), (_, b) in zip(m.named_parameters(), ref.named_parameters()):
            assert a.dtype == torch.float32
            tol = 2e-3 if name == "half" else 0.0
            assert (a - b).abs().max().item() <= tol * max(1.0, b.abs().max().item()), (name, k)
    bad = dict(sd)
    bad["feat_ext_conv1.0.weight"] = torch.zeros(8, 5, 3, 3)
    with pytest.raises(RuntimeError, match="feat_ext_conv1.0.weight has shape"):
        EMA_VFI(mid_channels=8).load_state_dict(bad)
    with pytest.raises(RuntimeError):  # strict: a reference checkpoint of another width does not load
        EMA_VFI(mid_channels=8).load_state_dict(synth.synthetic_state_dict(seed=5, mid_channels=16))


def test_device_code_has_no_packed_f32_operations(tmp_path):
    """on gfx950 a packed f32 VALU operation whose low half takes src1's high half reads zero in lanes
    48-63 beside the MFMAs of a kernel on another stream, so the build forbids the whole instruction class (csrc/Makefile, NOPK).  Guard the
    flag: disassemble every code object embedded in libemavfi.so and count v_pk_{add,mul,fma}_f32."""
    import re
    import shutil
    import subprocess
    llvm = "/opt/rocm/lib/llvm/bin"
    if not os.path.exists(f"{llvm}/clang-offload-bundler") or shutil.which("objcopy") is None:
        pytest.skip("ROCm binutils not available")
    so = lib.LIB_PATH
    fat = tmp_path / "fat.bin"
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", so, str(fat)])
    blob = fat.read_bytes()
    starts = [m.start() for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", blob)]
    assert len(starts) >= 4, "several bundles (one per translation unit with kernels) expected"
    kernels = packed = 0
    for i, a in enumerate(starts):
        part = tmp_path / f"bundle_{i}.bin"
        part.write_bytes(blob[a:starts[i + 1] if i + 1 < len(starts) else len(blob)])
        co = tmp_path / f"dev_{i}.co"
        subprocess.check_call([f"{llvm}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                               f"--input={part}", f"--output={co}"])
        dis = subprocess.run([f"{llvm}/llvm-objdump", "-d", str(co)], capture_output=True, text=True, check=True).stdout
        kernels += dis.count("s_endpgm")
        packed += len(re.findall(r"v_pk_(?:add|mul|fma)_f32", dis))
    assert kernels > 100, "disassembly looks empty"
    assert packed == 0, f"{packed} packed f32 VALU operations in the device code: was NOPK dropped from the build?"


def _device_disassembly(tmp_path):
    import re
    import shutil
    import subprocess
    llvm = "/opt/rocm/lib/llvm/bin"
    if not os.path.exists(f"{llvm}/clang-offload-bundler") or shutil.which("objcopy") is None:
        pytest.skip("ROCm binutils not available")
    fat = tmp_path / "fat.bin"
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib.LIB_PATH, str(fat)])
    blob = fat.read_bytes()
    starts = [m.start() for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", blob)]
    out = []
    for i, a in enumerate(starts):
        part = tmp_path / f"b_{i}.bin"
        part.write_bytes(blob[a:starts[i + 1] if i + 1 < len(starts) else len(blob)])
        co = tmp_path / f"d_{i}.co"
        subprocess.check_call([f"{llvm}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                               f"--input={part}", f"--output={co}"])
        out.append(subprocess.run([f"{llvm}/llvm-objdump", "-d", str(co)], capture_output=True, text=True, check=True).stdout)
    return out


def test_ring_kernels_issue_a_fixed_number_of_vmem_instructions_per_step(tmp_path):
    """csrc/conv_ring.inl retires its LDS-DMA ring with a COUNTED `s_waitcnt vmcnt(N)`: N is only right while every wave issues
    exactly 3 DMA + 2 store instructions per row step and nothing else that counts.  hipcc is free to
    break that silently (a branch over an all-inactive store, a spill, a hoisted load), so the shipped code objects are checked:
    in the row loop (the innermost loop around the counted wait) there are three global_load_lds, three buffer stores (the fused head
    kernel and the fused reconstruction tail: one) and no other vector-memory instruction, and N is stores + (3 + stores) (D - 2) (D = 3; the
    fused head kernel: D = 2)."""
    import re
    seen = {}
    for dis in _device_disassembly(tmp_path):
        for name, body in re.findall(r"<(_Z(?:19conv3x3_ring|23conv3x3_ringtail)_kernel\w+)>:\n(.*?)(?=\n\n|\Z)", dis, re.S):
            lines = body.splitlines()
            # the row loop's counted wait: the one in front of a barrier (hipcc's own waits for the weight loads of the kernel's
            # prologue look the same where N is small, but no barrier follows them)
            tops = [i for i, l in enumerate(lines) if re.search(r"s_waitcnt vmcnt\((\d+)\) lgkmcnt\(0\)", l) and "vmcnt(0)" not in l
                    and "s_barrier" in " ".join(lines[i:i + 4])]
            assert len(tops) == 1, (name, "one counted wait (the row loop's) expected", len(tops))
            n = int(re.search(r"vmcnt\((\d+)\)", lines[tops[0]]).group(1))
            # the row loop = the innermost loop (backward branch) around the counted wait.  (Until round 5 this test read the text from
            # the wait to the drain behind the loop; hipcc rotates the loop when the step ENDS with a store - the fused head kernel's
            # head_out - and lays that store out in front of the wait.)
            ins, loops = _loops_of(body)
            wait_addr = next(a for a, t in ins if t.startswith(lines[tops[0]].split("//")[0].strip()) and
                             any(t2 == "s_barrier" for _, t2 in ins[[x[0] for x in ins].index(a) + 1:[x[0] for x in ins].index(a) + 4]))
            row = min((l for l in loops if l[0] <= wait_addr <= l[1]), key=lambda l: l[1] - l[0])
            region = "\n".join(t for a, t in ins if row[0] <= a <= row[1])
            dma = len(re.findall(r"global_load_lds_dwordx4", region))
            stores = len(re.findall(r"buffer_store_dword", region))
            other = len(re.findall(r"\b(?:global_load_dword|global_store|buffer_load|flat_load|flat_store|scratch_)", region))
            m = re.search(r"ring_kernelI\w+?Lb(\d)ELb(\d)ELb(\d)E", name)   # <T, TAIL, HEAD, ALT>
            head, tail = bool(m) and m.group(2) == "1", "ringtail" in name   # (ringtail: ONE store - the frame's three planes leave together -, D = 3)
            assert (dma, stores, other) == (3, 1 if tail or head else 3, 0), (name, dma, stores, other)
            assert n == (5 if tail else 1 if head else 9), (name, n)
            seen[name] = n
    # bf16 and f16 instances of: 64 -> 64, the same storing the other 16-bit type (ALT), TAIL (67 -> 64), HEAD (+ flow head): 8;
    # reconstruction.1 + .2: bf16 and f16 x (round16, tanh head) = 8 (round 5: both became template arguments)
    assert len(seen) == 16, sorted(seen)


def _loops_of(body):
    """[(addr, text)] of a disassembled function and its backward-branch loops [(head address, branch address)]."""
    import re
    ins = []
    for line in body.splitlines():
        m = re.match(r"\s*(\S.*?)\s*//\s*([0-9A-Fa-f]{8,16}):", line)
        if m:
            ins.append((int(m.group(2), 16), m.group(1)))
    loops = []
    for addr, text in ins:
        m = re.match(r"s_c?branch\w* (\d+)$", text)     # target = next instruction + simm16 dwords
        if m:
            off = int(m.group(1))
            off = off - 65536 if off >= 32768 else off
            if off < 0:
                loops.append((addr + 4 + 4 * off, addr))
    return ins, loops


def test_ring_kernels_prime_their_rings_with_the_steady_state_pattern(tmp_path):
    """the counted `s_waitcnt vmcnt(N)` of the ring kernels is only right if the PRIMING loop in front of the row loop issues
    the same instruction pattern per iteration as a row step - NDMA global_load_lds + NSTORE (dropped) buffer stores - and nothing
    else that counts.  Checked on the shipped code objects, per loop (backward branches of the disassembly): the row loop (the
    innermost loop around the counted wait) and the priming loop (the last loop with LDS-DMA in front of it).  Round 4's two-layer
    kernel (conv_ring2.inl) is covered too: its A-waves issue 3 DMA and wait vmcnt(3) (D = 3, no stores), its B-waves 3 stores, in the
    same row loop.  The scheme relies on gfx9's single in-order vmcnt for loads, stores and LDS-DMA (a gfx10+ port with a separate
    vscnt must not reuse it: csrc/conv_ring.inl)."""
    import re
    seen = {}
    for dis in _device_disassembly(tmp_path):
        for name, body in re.findall(r"<(_Z(?:19conv3x3_ring|20conv3x3_ring2|23conv3x3_ringtail)_kernel\w+)>:\n(.*?)(?=\n\n|\Z)", dis, re.S):
            ins, loops = _loops_of(body)
            # (the counted wait stands in front of a barrier; hipcc's own waits for the prologue's weight loads do not)
            waits = [k for k, (a, t) in enumerate(ins) if re.match(r"s_waitcnt vmcnt\(\d+\) lgkmcnt\(0\)", t) and "vmcnt(0)" not in t]
            if len(waits) > 1:
                waits = [k for k in waits if any(t2 == "s_barrier" for _, t2 in ins[k + 1:k + 4])]
            waits = [ins[k][0] for k in waits]
            assert len(waits) == 1, (name, len(waits))
            n = int(re.search(r"vmcnt\((\d+)\)", next(t for a, t in ins if a == waits[0])).group(1))
            def bars(l):
                return sum(t == "s_barrier" for a, t in ins if l[0] <= a <= l[1])
            # the row loop: the widest loop around the counted wait with ONE s_barrier (the item loop around it has more; conv_ring2's row
            # loop has one back edge per role - A-waves: 3 DMA, B-waves: 3 stores - so its narrower loops cover one role only)
            row = max((l for l in loops if l[0] <= waits[0] <= l[1] and bars(l) == 1), key=lambda l: l[1] - l[0])
            # the priming loop: the innermost loop with LDS-DMA in front of the row loop
            prime = min((l for l in loops if l[1] < row[0] and any("global_load_lds" in t for a, t in ins if l[0] <= a <= l[1])), key=lambda l: l[1] - l[0])

            def count(loop):
                txt = [t for a, t in ins if loop[0] <= a <= loop[1]]
                dma = sum("global_load_lds_dwordx4" in t for t in txt)
                st = sum(t.startswith("buffer_store_dword") for t in txt)
                other = sum(bool(re.match(r"(global_load_dword|global_store|buffer_load|flat_load|flat_store|scratch_)", t)) for t in txt)
                return dma, st, other
            m = re.search(r"ring_kernelI\w+?Lb(\d)ELb(\d)ELb(\d)E", name)
            ring2, head = "ring2" in name, bool(m) and m.group(2) == "1"
            want_row = (3, 3, 0) if ring2 else (3, 1 if "ringtail" in name or head else 3, 0)
            want_prime = (3, 0, 0) if ring2 else want_row
            assert count(row) == want_row, (name, "row loop", count(row))
            if ring2:   # both roles' contractions are in it: 2 x 36 MFMAs
                assert sum("v_mfma" in t for a, t in ins if row[0] <= a <= row[1]) == 72, name
            assert count(prime) == want_prime, (name, "priming loop", count(prime))
            assert n == (3 if ring2 else 5 if "ringtail" in name else 1 if head else 9), (name, n)
            seen[name] = n
    # ring: 8 instances (bf16 / f16 x {64 -> 64, ALT, TAIL, HEAD}); ringtail: 8 (bf16 / f16 x round16 x tanh head); ring2: 4 (bf16 / f16 x ALT)
    assert len(seen) == 20, sorted(seen)


def _vregs(text):
    """VGPR numbers an instruction's operands name (v7, v[4:7])."""
    import re
    regs = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", text):
        regs.update(range(int(a), int(b) + 1))
    regs.update(int(a) for a in re.findall(r"\bv(\d+)\b", text))
    return regs


def _touches_of_loads_in_flight(txt):
    """Instructions (other than the counted waits) that read or write a VGPR while an inline-assembly weight load into it is still in
    flight, in a straight-line reading of the function: VMEM retires in issue order, `s_waitcnt vmcnt(k)` leaves the youngest k."""
    import re
    fifo, bad = [], []
    for t in txt:
        m = re.match(r"s_waitcnt\s+(?:.*?)vmcnt\((\d+)\)", t)
        if m:
            k = int(m.group(1))
            fifo = fifo[len(fifo) - k:] if 0 < k < len(fifo) else ([] if k == 0 else fifo)
            continue
        if re.match(r"(s_cbranch|s_branch|s_barrier|s_endpgm)", t):
            continue
        pend = set().union(*fifo) if fifo else set()
        m = re.match(r"global_load_dwordx4 (v\[\d+:\d+\]), (v\d+), s\[", t)       # the asm weight loads: saddr form
        if m:
            if _vregs(m.group(2)) & pend:
                bad.append(t)
            fifo.append(_vregs(m.group(1)))
            continue
        if re.match(r"(global_|buffer_|flat_|scratch_)", t):
            fifo.append(set())
        if _vregs(t) & pend:
            bad.append(t)
    return bad


def test_wreg_kernels_count_their_waits_exactly(tmp_path):
    """csrc/conv_wreg.inl (context_encoding.1 / .2) streams its weights with inline-assembly global_load_dwordx4 and its input tile with
    inline-assembly LDS-DMA, both behind COUNTED `s_waitcnt vmcnt(N)` that hipcc knows nothing about: "step s's two fragments have
    landed" is N = 2 P, or 2 P + NI in the first P steps of a chunk (the next chunk's NI DMA instructions are younger than those
    steps' loads).  The counts only hold while the k loop contains exactly those instructions - a spill (scratch), a flat access, an
    ordinary load or a store inside it would shift every count, silently.  And hipcc believes an asm load's result is in its registers
    when the statement ends: a copy, a spill or a reuse between the load and its wait reads (or clobbers) data still in flight (an
    experiment with the wait in two arms of a branch got its "+v" registers merged by copies IN FRONT of the wait: sporadic wrong
    weights on the GPU), and an "s" operand that arrives through v_readfirstlane gets none of the wait states a VMEM instruction needs
    behind a VALU write of an SGPR (stale base address: a memory fault in another experiment).  Checked on the shipped code objects
    (S = 1: P = 3, NI = 8, 36 steps per chunk; S = 2: P = 4, NI = 10, 18 steps), together with the register budget that lets two
    workgroups share a CU."""
    import re
    seen = {}
    for dis in _device_disassembly(tmp_path):
        for name, body in re.findall(r"<(_Z19conv3x3_wreg_kernel\w+)>:\n(.*?)(?=\n\n|\Z)", dis, re.S):
            S = int(re.search(r"Li(\d)EEv", name).group(1))
            P, NI, SPC = (3, 8, 36) if S == 1 else (4, 10, 18)
            ins, _ = _loops_of(body)
            txt = [t for _, t in ins]
            assert not _touches_of_loads_in_flight(txt), (name, _touches_of_loads_in_flight(txt)[:4])
            first_bar = txt.index("s_barrier")
            last_mfma = max(i for i, t in enumerate(txt) if t.startswith("v_mfma"))
            loop = txt[first_bar:last_mfma + 1]
            assert sum(t.startswith("v_mfma_f32_32x32x16") for t in loop) == 2 * SPC * 8, name            # two copies of the chunk body (not-last, last)
            assert sum(t.startswith("global_load_lds_dwordx4") for t in loop) == NI, name                # the next chunk's DMA, in the not-last copy only
            wl = [t for t in loop if t.startswith("global_load_dwordx4")]
            assert len(wl) == 2 * (2 * SPC - P) and all(re.search(r", s\[\d+:\d+\]", t) for t in wl), (name, len(wl))   # weights: saddr form, two per step
            other = [t for t in loop if re.match(r"(global_load_dword\b|global_load_ubyte|global_load_ushort|global_store|buffer_|flat_|scratch_)", t)]
            assert not other, (name, other[:4])
            waits = [int(m.group(1)) for t in loop for m in [re.match(r"s_waitcnt vmcnt\((\d+)\)", t)] if m]
            want = [2 * P + NI] * P + [2 * P] * (SPC - P) + [2 * P]                                     # not-last chunk: its steps, then the boundary wait
            want += [2 * min(P, SPC - 1 - sc) for sc in range(SPC)]                                     # last chunk: the prefetch runs dry
            # hipcc adds its own waits for the bias loads (behind the first barrier, and a vmcnt(0) on the one-chunk entry path between the
            # two copies; they only strengthen): compare the non-zero counts' tail, and the very last wait (the last step: nothing younger)
            nz, want_nz = [w for w in waits if w], [w for w in want if w]
            assert nz[-len(want_nz):] == want_nz and waits[-1] == 0, (name, nz[-len(want_nz):][:8], want_nz[:8], waits[-3:])
            assert loop.count("s_barrier") == 2, name                                                     # the prologue's and the chunk boundary's
            # the weight loads' base is an SGPR pair SALU / SMEM wrote: no v_readfirstlane into it anywhere near (VALU write of an SGPR ->
            # VMEM read needs wait states nobody inserts in front of an asm string)
            bases = {m.group(1) for t in wl for m in [re.search(r", s\[(\d+):\d+\]", t)]}
            vrf = {m.group(1) for t in loop for m in [re.match(r"v_readfirstlane_b32 s(\d+),", t)] if m}
            assert not ({int(b) for b in bases} | {int(b) + 1 for b in bases}) & {int(v) for v in vrf}, (name, bases, vrf)
            seen[name] = S
    assert sorted(seen.values()) == [1, 1, 2, 2], sorted(seen)     # bf16 and f16, stride 1 and 2


# Every kernel family of the library that moves data with LDS-DMA (global_load_lds: hand-written asm or the compiler's builtin - the
# disassembly cannot tell them apart) or with loads hipcc does not know of.  "asm": the DMA / loads are inline assembly retired by COUNTED
# waits the compiler cannot check - the named test pins their instruction counts on the shipped code objects.  "builtin": the compiler
# sees the DMA and drains it itself; only the generic barrier rule below applies.  A NEW family with LDS-DMA fails the enumeration
# test until it is listed here - and an "asm" entry must name a test that exists.
_VMEM_FAMILIES = {
    "conv3x3_ring_kernel": ("asm", "test_ring_kernels_issue_a_fixed_number_of_vmem_instructions_per_step"),
    "conv3x3_ringtail_kernel": ("asm", "test_ring_kernels_issue_a_fixed_number_of_vmem_instructions_per_step"),
    "conv3x3_ring2_kernel": ("asm", "test_ring_kernels_prime_their_rings_with_the_steady_state_pattern"),
    "conv3x3_wreg_kernel": ("asm", "test_wreg_kernels_count_their_waits_exactly"),
    "conv3x3_s2ring_kernel": ("builtin", None), "conv3x3_kernel": ("builtin", None), "conv3x3_persist_kernel": ("builtin", None),
    "conv3x3_persist16_kernel": ("builtin", None), "conv_light_kernel": ("builtin", None), "deform_pack3_kernel": ("builtin", None),
    "deform_f32w_kernel": ("builtin", None), "warp_tiled_kernel": ("builtin", None),
}


def test_every_kernel_with_lds_dma_is_listed_and_waits_before_its_barrier(tmp_path):
    """One enumeration over EVERY kernel symbol of the shipped code objects: (1) a kernel that contains
    global_load_lds belongs to a family listed in _VMEM_FAMILIES, and every listed family exists - so an inline-assembly VMEM kernel cannot
    ship without somebody deciding which code-object test pins its counted waits; (2) the generic rule of an LDS-DMA producer: between a
    global_load_lds and the NEXT s_barrier (the point where other waves start reading what it wrote) there is an `s_waitcnt` with a vmcnt
    field - counted or zero - in the ten instructions in front of that barrier.  hipcc places vmcnt(0) there for its builtin today
    (warp_tiled_kernel now also says so itself), the asm kernels place their counted wait; a compiler that moved the wait behind the
    barrier would otherwise produce silently wrong windows; (3) no family of the "asm" kind carries flat_ instructions anywhere (they
    count in vmcnt AND lgkmcnt: DESIGN 3.2b, the lost address space of an indexed LDS base array) or reads an SGPR written by
    v_readfirstlane as the m0 / saddr operand of its DMA inside a loop."""
    import re
    seen = {}
    for dis in _device_disassembly(tmp_path):
        for name, body in re.findall(r"<(\w+)>:\n(.*?)(?=\n\n|\Z)", dis, re.S):
            if "s_endpgm" not in body or "global_load_lds" not in body:
                continue
            m = re.match(r"_Z\d+(\w+?_kernel)", name)
            fam = m.group(1) if m else name
            assert fam in _VMEM_FAMILIES, f"{name}: a kernel with LDS-DMA that tests/test_cabi_cpu.py::_VMEM_FAMILIES does not list"
            kind, test = _VMEM_FAMILIES[fam]
            ins, loops = _loops_of(body)
            txt = [t for _, t in ins]
            pending = False
            for i, t in enumerate(txt):
                if t.startswith("global_load_lds"):
                    pending = True
                elif t == "s_barrier" and pending:
                    near = txt[max(0, i - 10):i]   # (conv_ring2: the counted wait sits in the A-role arm of a branch that ends right in front of the barrier)
                    assert any(re.match(r"s_waitcnt\b.*vmcnt\(\d+\)", w) for w in near), (name, "barrier behind LDS-DMA without a vmcnt wait", near)
                    pending = False
            if kind == "asm":
                assert test in globals(), (fam, test)
                assert not any(t.startswith("flat_") for t in txt), (name, "flat access in a counted-wait kernel")
                vrf = {int(mm.group(1)) for t in txt for mm in [re.match(r"v_readfirstlane_b32 s(\d+),", t)] if mm}
                for (a0, a1) in loops:
                    inloop = [t for a, t in ins if a0 <= a <= a1]
                    for j, t in enumerate(inloop):
                        if t.startswith("global_load_lds"):
                            mm = re.match(r"s_mov_b32 m0, s(\d+)", next((w for w in reversed(inloop[:j]) if w.startswith("s_mov_b32 m0")), ""))
                            # (m0 fed from an SGPR: that SGPR must not be a v_readfirstlane result produced inside the same loop)
                            if mm and int(mm.group(1)) in vrf:
                                assert not any(re.match(rf"v_readfirstlane_b32 s{mm.group(1)},", w) for w in inloop), (name, "m0 from v_readfirstlane inside the loop")
            seen.setdefault(fam, 0)
            seen[fam] += 1
    assert set(seen) == set(_VMEM_FAMILIES), (sorted(set(_VMEM_FAMILIES) - set(seen)), sorted(set(seen) - set(_VMEM_FAMILIES)))


def test_packed_cache_file_carries_a_checksum(tmp_path):
    """the on-disk packed-weight cache trusted any file of the right size.  The file is now blob + sha256(blob);
    a flipped byte, a truncated file or a file of the old format is ignored (the caller re-packs and overwrites it)."""
    import torch
    from emavfi import EMA_VFI
    blob = torch.arange(4096, dtype=torch.int64).to(torch.uint8)
    path = str(tmp_path / "packed_test.bin")
    EMA_VFI._cache_write(path, blob)
    assert os.path.getsize(path) == 4096 + 32
    back = EMA_VFI._cache_read(path, 4096, "cpu")
    assert back is not None and torch.equal(back, blob)
    raw = bytearray(open(path, "rb").read())
    raw[100] ^= 1
    open(path, "wb").write(bytes(raw))
    assert EMA_VFI._cache_read(path, 4096, "cpu") is None                 # bit rot
    open(path, "wb").write(bytes(raw[:4096]))
    assert EMA_VFI._cache_read(path, 4096, "cpu") is None                 # old format / truncated
    assert EMA_VFI._cache_read(path, 4095, "cpu") is None                 # other size
    assert EMA_VFI._cache_read(str(tmp_path / "missing.bin"), 4096, "cpu") is None


def _host_blob(L, mid, dt, tag=None, version=None, dtype_field=None):
    """A blob in HOST memory built from include/emavfi.h's description of the header alone (no kernel runs here)."""
    import numpy as np
    total = L.emavfi_packed_bytes(3, mid, 3, dt)
    blob = np.zeros(total, dtype=np.uint8)
    rng = np.random.default_rng(7)
    blob[256:] = rng.integers(0, 256, total - 256, dtype=np.uint8)
    words = blob[256:].view(np.uint32).astype(np.uint64)
    with np.errstate(over="ignore"):
        checksum = ((words + np.uint64(0x9E3779B9)) * (np.uint64(2) * np.arange(words.size, dtype=np.uint64) + np.uint64(1))).sum(dtype=np.uint64)
    hdr = np.zeros(16, dtype=np.uint32)
    hdr[0:2] = np.frombuffer(b"EMAVFIPK", dtype=np.uint32)
    hdr[2] = L.emavfi_version() if version is None else version
    hdr[3] = 256
    hdr[4:8] = (3, mid, 3, dt if dtype_field is None else dtype_field)
    hdr[8] = L.emavfi_layout_tag() if tag is None else tag
    blob[:64] = hdr.view(np.uint8)
    blob[40:48] = np.frombuffer(np.uint64(total).tobytes(), dtype=np.uint8)
    blob[48:56] = np.frombuffer(np.uint64(checksum).tobytes(), dtype=np.uint8)
    return blob


def test_packed_blob_is_self_describing():
    """VERDICT r3 item 8 / ADVICE r3: the packed blob starts with a header (magic, library version, model, dtype, layout-switch tag,
    size, payload checksum) and emavfi_packed_check names what is wrong with a foreign one.  Host memory here: the check reads
    device memory through a copy and anything else in place."""
    L = lib.load()
    assert L.emavfi_layout_tag() == 0, "the suite runs without layout switches in the environment"
    assert lib.layout_switches() == "layout_tag=0"

    def check(blob, mid=8, dt=lib.BF16, nbytes=None):
        rc = L.emavfi_packed_check(3, mid, 3, dt, blob.ctypes.data, blob.size if nbytes is None else nbytes)
        return rc, lib.last_error()

    good = _host_blob(L, 8, lib.BF16)
    assert check(good) == (0, lib.last_error())
    flipped = good.copy()
    flipped[5000] ^= 0x10
    rc, msg = check(flipped)
    assert rc == -1 and "checksum" in msg
    rc, msg = check(good, dt=lib.F16)                       # same size, other dtype: only the header can tell
    assert rc == -1 and "dtype" in msg
    rc, msg = check(_host_blob(L, 8, lib.BF16, tag=2))       # packed under EMAVFI_CONV_RING=0
    assert rc == -1 and "layout switches" in msg
    rc, msg = check(_host_blob(L, 8, lib.BF16, version=300))
    assert rc == -1 and "version 300" in msg
    rc, msg = check(_host_blob(L, 16, lib.BF16), mid=16)
    assert rc == 0
    rc, msg = check(_host_blob(L, 16, lib.BF16)[:L.emavfi_packed_bytes(3, 16, 3, lib.BF16)], mid=8)   # a bigger model's blob
    assert rc == -1 and "EMA_VFI(3, 16, 3)" in msg
    headerless = good.copy()
    headerless[:8] = 0
    rc, msg = check(headerless)
    assert rc == -1 and "EMAVFIPK" in msg
    rc, msg = check(good, nbytes=good.size - 1)
    assert rc == -1 and "bytes" in msg
    assert L.emavfi_packed_check(3, 8, 3, lib.BF16, None, 100) == -1
    # the forward refuses a buffer shorter than the model needs before it touches anything
    fake = ctypes.c_void_p(256)
    rc = L.emavfi_forward(3, 64, 3, fake, 1000, fake, fake, fake, fake, 1 << 40, 1, 64, 64, lib.BF16, None, None)
    assert rc == -1 and "packed blob has 1000 bytes" in lib.last_error()


def test_mdcn_entry_validates_its_arguments():
    """emavfi_mdcn (one ModulatedDeformConvPack.forward, routed as a block of the forward): host-side queries and guards."""
    L = lib.load()
    assert L.emavfi_mdcn_workspace_bytes(1, 67, 32, 32, lib.BF16, 0) > 0
    assert L.emavfi_mdcn_workspace_bytes(1, 67, 32, 32, lib.BF16, lib.MDCN_SPLIT_TAIL) > L.emavfi_mdcn_workspace_bytes(1, 67, 32, 32, lib.BF16, 0)
    assert L.emavfi_mdcn_workspace_bytes(1, 67, 32, 32, lib.AMP16, 0) > L.emavfi_mdcn_workspace_bytes(1, 67, 32, 32, lib.F16, 0)
    assert L.emavfi_mdcn_workspace_bytes(1, 11, 32, 32, lib.F32, 0) > 0
    assert L.emavfi_mdcn_workspace_bytes(1, 66, 32, 32, lib.F32, 0) == 0 and "mid_channels + 3" in lib.last_error()
    assert L.emavfi_mdcn_workspace_bytes(1, 67, 32, 32, lib.F32, lib.MDCN_IN_F16) == 0 and "f16 hand-off" in lib.last_error()
    assert L.emavfi_mdcn_workspace_bytes(1, 67, 32, 32, lib.F16, lib.MDCN_OUT_F16) == 0      # bf16 models only
    assert L.emavfi_mdcn_workspace_bytes(1, 11, 32, 32, lib.BF16, lib.MDCN_SPLIT_TAIL) == 0   # no one-launch kernel at that width
    assert L.emavfi_mdcn_workspace_bytes(1, 67, 32, 32, lib.BF16, 8) == 0 and "flag" in lib.last_error()
    fake = ctypes.c_void_p(256)
    assert L.emavfi_mdcn(None, fake, fake, fake, fake, fake, 1, 67, 8, 8, lib.BF16, 0, fake, 0, None) == -1
    assert L.emavfi_mdcn(fake, fake, fake, fake, fake, fake, 1, 67, 8, 8, lib.BF16, 0, fake, 16, None) == -3


def test_switch_word_is_latched_and_settable():
    """the launch-sequence switches are read from the environment once; tests flip them through emavfi_debug_switches."""
    L = lib.load()
    old = lib.debug_switches()
    try:
        assert L.emavfi_forward_launches(3, 64, 3, 1, 64, 64, lib.BF16, None, 0, None, None, 0) == 17 - RING2
        lib.debug_switches(~lib.SW_NO_HEAD, lib.SW_NO_HEAD)
        os.environ["EMAVFI_CONV_HEAD"] = "1"            # the environment is not consulted again
        try:
            assert lib.debug_switches() & lib.SW_NO_HEAD
            n = L.emavfi_forward_launches(3, 64, 3, 1, 64, 64, lib.BF16, None, 0, None, None, 0)
        finally:
            del os.environ["EMAVFI_CONV_HEAD"]
        assert n > 17 - RING2
    finally:
        lib.debug_switches(0, old)
    assert lib.debug_switches() == old


def test_host_side_runs_clean_under_asan_ubsan():
    """SURVEY.md section 5 (sanitizers) / VERDICT r3 item 9: `make asan` builds libemavfi_asan.so with AddressSanitizer +
    UndefinedBehaviorSanitizer in every HOST pass (the gfx950 device code stays uninstrumented: GPU ASan / XNACK do not exist on this
    pool) and tests/host/host_check, which drives every entry that works or refuses on the host: plan building for every model /
    dtype / block count, workspace carving, launch enumeration with exact and short buffers, every argument guard, the blob header
    check on host memory, the switch word.  Any report aborts the program (-fno-sanitize-recover).  It found one bug when it was
    introduced: pointer arithmetic on the enumeration pass's null blob pointer."""
    import shutil
    import subprocess
    llvm = "/opt/rocm/lib/llvm/bin/clang"
    if not os.path.exists(llvm) or shutil.which("make") is None:
        pytest.skip("ROCm clang not available")
    rt = subprocess.run([llvm, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(rt) or not os.path.exists(rt):
        pytest.skip("no shared ASan runtime in this toolchain")
    csrc = os.path.join(ROOT, "video-frame-interpolation_amd", "csrc")
    r = subprocess.run(["make", "-C", csrc, "asan", "-j", str(min(8, os.cpu_count() or 1))], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    exe = os.path.join(ROOT, "build", "csrc_asan", "host_check")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1",
               LD_LIBRARY_PATH=os.path.dirname(rt) + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "host_check: ok" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-3000:]
