"""CPU-side ISA screen of the production kernels (no GPU needed: hipcc cross-compiles gfx950).

Round 5 found two silent regressions in the SHIPPED code objects that no numerical test can see: the O / FC2 instantiation of
`gemm16_h256_kernel` kept 12 VGPRs in scratch around the middle loop of every tile (an ablation path had made its stream registers
live across it: 3 scratch stores + 4 reloads per tile, each reload behind `s_waitcnt vmcnt(0)`), and every instantiation spilled 11-21
SGPRs to VGPR lanes for run-time ablation bits.  This test compiles the two kernel files of the headline's encoder path to assembly and
reads the code-object metadata: no scratch, no spilled VGPR in any production instantiation, SGPR spills bounded, two waves per SIMD."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "iisan_amd", "csrc")


def _kernel_meta(src, tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = tmp_path / (os.path.basename(src) + ".s")
    subprocess.check_call([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", src, "-o", str(out)],
                          stderr=subprocess.DEVNULL)
    txt = out.read_text()
    meta = {}
    for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.wavefront_size", txt, re.S):
        get = lambda k: int(re.search(k + r":\s*(\d+)", m.group(2)).group(1))
        meta[m.group(1)] = dict(vgpr=get(r"\.vgpr_count"), vgpr_spill=get(r"\.vgpr_spill_count"), sgpr_spill=get(r"\.sgpr_spill_count"),
                                scratch=get(r"\.private_segment_fixed_size"))
    return meta, txt


def test_encoder_gemm_instantiations_have_no_scratch_and_no_spilled_vgpr(tmp_path):
    meta, txt = _kernel_meta(os.path.join(CSRC, "gemm16_h256.hip"), tmp_path)
    kernels = {k: v for k, v in meta.items() if "gemm16_h256_kernel" in k}
    assert len(kernels) >= 11                         # seven fp16 + four bf16 instantiations
    for name, m in kernels.items():
        assert m["scratch"] == 0 and m["vgpr_spill"] == 0, (name, m)
        assert m["vgpr"] <= 256, (name, m)            # two 256-register waves per SIMD (launch_bounds(512, 2))
        assert m["sgpr_spill"] <= 8, (name, m)        # (measured 0 - 5 with the ablation bits compiled out; 11 - 21 with them in)
    # the product build must not carry the ablation bits
    assert "GEMM16_DEBUG_BITS" not in open(os.path.join(CSRC, "Makefile")).read()


def test_attention_instantiations_have_no_scratch_and_keep_two_waves_per_simd(tmp_path):
    meta, txt = _kernel_meta(os.path.join(CSRC, "attn16.hip"), tmp_path)
    kernels = {k: v for k, v in meta.items() if "attention16_kernel" in k}
    assert len(kernels) >= 10
    for name, m in kernels.items():
        assert m["scratch"] == 0 and m["vgpr_spill"] == 0, (name, m)
        assert m["vgpr"] <= 256, (name, m)
    # round 5: the per-key limit must be a compiler-visible instruction (an inline-asm v_min read its MFMA result too early once the
    # branches around it were gone: the hazard recognizer does not look inside inline asm)
    src = open(os.path.join(CSRC, "attn16.hip")).read()
    assert 'asm("v_min_f32' not in src and "__builtin_amdgcn_fmed3f" in src
