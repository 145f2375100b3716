"""Static ISA check of the token-major GEMM (tools/check_tn_isa.py): its `ds_read_b64_tr_b16` fragment reads are inline asm the compiler cannot see as
asynchronous loads, so nothing may touch their destination registers before the next `s_waitcnt lgkmcnt(0)`.  Checked on the compiled kernel itself
(the build container has hipcc; the GPU tests additionally hold the shipped binary bit-identical to the transposed form)."""
import shutil
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))
import check_tn_isa as C  # noqa: E402

K = "_ZN4amds16gemm_4w16_kernelIDF16bLi4ELi6ELi6ELb1ELi0ELi0ELb1EEEvPKT_lS3_liiiNS_7EpiArgsEii:"


def test_checker_flags_an_early_use_and_accepts_a_waited_one():
    good = f"""{K}
	ds_read_b64_tr_b16 v[10:11], v3
	ds_read_b64_tr_b16 v[12:13], v3 offset:2048
	v_add_u32_e32 v3, 64, v3
	s_waitcnt lgkmcnt(0)
	v_mfma_f32_16x16x32_bf16 a[0:3], v[20:23], v[10:13], a[0:3]
	s_endpgm
"""
    assert C.check(good) == []
    moved = good.replace("v_add_u32_e32 v3, 64, v3", "v_mov_b32_e32 v40, v11")
    assert len(C.check(moved)) == 1 and "v_mov_b32_e32" in C.check(moved)[0]
    early_mfma = good.replace("	s_waitcnt lgkmcnt(0)\n", "")
    assert len(C.check(early_mfma)) == 1 and "v_mfma" in C.check(early_mfma)[0]
    addr = good.replace("ds_read_b64_tr_b16 v[12:13], v3 offset:2048", "ds_read_b64_tr_b16 v[12:13], v10 offset:2048")
    assert any("address register" in e for e in C.check(addr))
    assert C.check("_ZN4amds16gemm_4w16_kernelIDF16bLi4ELi6ELi6ELb1ELi0ELi0ELb0EEEvPKT_lS3_liiiNS_7EpiArgsEii:\n\ts_endpgm\n") != []      # no TN kernel at all


@pytest.mark.skipif(not Path(C.HIPCC).exists() or shutil.which("make") is None, reason="needs hipcc")
@pytest.mark.parametrize("name", ["gemm_bf16", "gemm_f16"])
def test_compiled_token_major_kernel_respects_its_own_waits(name):
    """Both instantiations, compiled with the flags the Makefile itself uses (read out of `make -pn`), the way its `.tn_ok` build step does."""
    import subprocess
    import tempfile
    db = subprocess.run(["make", "-C", str(ROOT), "-pn", "__no_such_target__"], capture_output=True, text=True).stdout
    flags = next(line.split(":=", 1)[1].split() for line in db.splitlines() if line.startswith("HIPFLAGS :="))
    assert "--offload-arch=gfx950" in flags and "-O3" in flags
    assert f"{name}" in next(line for line in db.splitlines() if line.startswith("TN_SRCS :=")), "the Makefile no longer checks this file at build time"
    f = ROOT / "stamp_amd" / "csrc" / f"{name}.hip"
    with tempfile.TemporaryDirectory() as td:
        out = Path(td) / "k.s"
        subprocess.run([C.HIPCC, *flags, "--cuda-device-only", "-S", "-o", str(out), str(f)],
                       check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=ROOT)
        errs = C.check(out.read_text())
    assert errs == [], errs[:5]
