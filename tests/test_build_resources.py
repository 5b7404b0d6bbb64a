"""Build-time guard (CPU, hipcc cross-compiles without a GPU): register allocation of every kernel.

A change that let hipcc hoist the in-kernel policy's weight loads out of the step loop once cost 7x
without a single wrong bit -- 4 878 SGPR spills to VGPR lanes in one kernel, nothing in any parity
test.  The code-object metadata shows that kind of accident at once."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gym-acas2d_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="needs hipcc")
@pytest.mark.parametrize("unit", ("acas2d_f32.hip", "acas2d_f64.hip"))
def test_no_kernel_spills_or_uses_scratch(unit, tmp_path):
    asm = tmp_path / (unit + ".s")
    subprocess.run([HIPCC if os.path.exists(HIPCC) else "hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17",
                    "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-ffp-contract=off", "-fno-slp-vectorize", "-mllvm",
                    "-amdgpu-kernarg-preload-count=8", "-S",
                    "--cuda-device-only", "-o", str(asm), os.path.join(CSRC, unit)], check=True, capture_output=True)
    text = asm.read_text()
    kernels = re.findall(r"\.name:\s+(\S+)\n(.*?)\.wavefront_size", text, re.S)
    assert len(kernels) >= 40
    field = lambda body, k: int(re.search(r"\.%s:\s+(\d+)" % k, body).group(1))  # noqa: E731
    for name, body in kernels:
        assert field(body, "vgpr_spill_count") == 0, name
        if field(body, "private_segment_fixed_size") != 0:                      # no scratch memory at all ...
            # ... except a small frame the register allocator reserved and then did not need (SGPR pressure: the
            # generic-walk reset_kernel, the float64 actor-critic rollouts -- not the per-step kernels): no instruction
            # may touch it.  One known exception, NON-default work shapes with two aircraft per lane (ACAS2D_SHAPE="2,4" /
            # "2,32", shape-sweep tests only): hipcc gathers four pinned launch constants into a vector there and pulls
            # operand pairs out of it through a 16-byte stack slot (4 instructions).
            code = text[text.index("\n" + name + ":"):text.index(".end_amdhsa_kernel", text.index("\n" + name + ":"))]
            touched = len(re.findall(r"\b(scratch_|buffer_)(load|store)", code))
            if "step_kernelIfLi2ELi32E" in name or "step_kernelIfLi2ELi4E" in name:
                assert field(body, "private_segment_fixed_size") <= 32 and touched <= 4, (name, touched)
            else:
                assert field(body, "private_segment_fixed_size") <= 128 and touched == 0, (name, touched)
            assert "step_kernelIfLi4ELi2ELb1ELb1ELb1ELb0ELb0E" not in name       # the headline kernel: no frame at all
        assert field(body, "sgpr_spill_count") < 400, (name, field(body, "sgpr_spill_count"))
        if "step_kernelIfLi4ELi2ELb1ELb1ELb1ELb0ELb0E" in name:                 # the headline kernel: >= 4 waves / SIMD
            assert field(body, "vgpr_count") <= 128, field(body, "vgpr_count")


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="needs hipcc")
def test_ppo_update_kernels_stay_in_registers(tmp_path):
    """csrc/acas2d_ppo.hip keeps three 64-entry per-lane vectors in registers (h1, dh1, a gradient row): a spill to
    scratch there would cost far more than any wrong bit would show."""
    asm = tmp_path / "acas2d_ppo.s"
    subprocess.run([HIPCC if os.path.exists(HIPCC) else "hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17",
                    "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-S", "--cuda-device-only", "-o", str(asm),
                    os.path.join(CSRC, "acas2d_ppo.hip")], check=True, capture_output=True)
    kernels = re.findall(r"\.name:\s+(\S+)\n(.*?)\.wavefront_size", asm.read_text(), re.S)
    assert len(kernels) == 6                                     # five observation widths + the apply kernel
    field = lambda body, k: int(re.search(r"\.%s:\s+(\d+)" % k, body).group(1))  # noqa: E731
    for name, body in kernels:
        assert field(body, "vgpr_spill_count") == 0 and field(body, "sgpr_spill_count") == 0, name
        assert field(body, "private_segment_fixed_size") == 0 and field(body, "vgpr_count") <= 256, name
