"""Static check of the token-major GEMM (gemm_4w16.h TN, kernel id 15).  Its MFMA fragments are read with `ds_read_b64_tr_b16` as INLINE ASM (the compiler's
builtin costs an s_waitcnt vmcnt(0) per read pair, profiles/r04_wgrad_tn.txt), so the compiler does not know the destination registers are filled
asynchronously: the kernel is only correct as long as NOTHING touches a destination register between the read and the next `s_waitcnt lgkmcnt(0)` -- no move,
no MFMA.  This script compiles the kernel to assembly and checks exactly that on the instruction stream of every TN kernel (linear scan; the K loop is
straight-line code).   python tools/check_tn_isa.py [file.hip ...]   -> exit status 0 / 1
                       python tools/check_tn_isa.py --asm file.s [...]  -> the same check on assembly the BUILD produced (the Makefile runs this on the -S
                       output of the build's own HIPFLAGS for gemm_bf16.hip and gemm_f16.hip and fails the build on a violation)"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
HIPCC = "/opt/rocm/bin/hipcc"
REG = re.compile(r"\b[va]\[(\d+):(\d+)\]|\bv(\d+)\b")


def vregs(text: str) -> set[int]:
    out: set[int] = set()
    for m in REG.finditer(text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        elif text[m.start()] == "v":
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def check(asm: str) -> list[str]:
    errors, kernel, pending = [], None, {}
    n_tr = 0
    for ln, line in enumerate(asm.splitlines(), 1):
        s = line.strip()
        m = re.match(r"^(_ZN4amds16gemm_4w16_kernel\S*?Lb1EEEv\S*):", s)        # ... TN = true (last template argument)
        if m:
            kernel, pending = m.group(1), {}
            continue
        if kernel is None or not s or s.startswith((";", ".", "//")):
            continue
        if s.startswith("s_endpgm"):
            kernel = None
            continue
        op = s.split()[0]
        if op == "s_waitcnt" and "lgkmcnt(0)" in s:
            pending = {}
            continue
        if op == "ds_read_b64_tr_b16":
            n_tr += 1
            dst, rest = s[len(op):].split(",", 1)
            used = vregs(rest.split("offset")[0])
            bad = used & set(pending)
            if bad:
                errors.append(f"{kernel[:60]}... line {ln}: address register(s) {sorted(bad)} of a transpose read are still in flight: {s}")
            for r in vregs(dst):
                pending[r] = ln
            continue
        if pending:
            bad = vregs(s.split(";")[0]) & set(pending)
            if bad:
                errors.append(f"{kernel[:60]}... line {ln}: `{s.split(';')[0].strip()}` touches v{sorted(bad)} (transpose read at line {pending[min(bad)]}) before s_waitcnt lgkmcnt(0)")
    if n_tr == 0:
        errors.append("no ds_read_b64_tr_b16 found: is the TN kernel still instantiated?")
    return errors


def main() -> int:
    if len(sys.argv) > 1 and sys.argv[1] == "--asm":
        rc = 0
        for f in sys.argv[2:]:
            errs = check(Path(f).read_text())
            print(f"{Path(f).name}: {'token-major fragment reads ok' if not errs else str(len(errs)) + ' violation(s)'}")
            for e in errs[:10]:
                print("   ", e)
            rc |= bool(errs)
        return rc
    files = [Path(f) for f in sys.argv[1:]] or [ROOT / "stamp_amd" / "csrc" / "gemm_bf16.hip", ROOT / "stamp_amd" / "csrc" / "gemm_f16.hip"]
    rc = 0
    for f in files:
        with tempfile.TemporaryDirectory() as td:
            out = Path(td) / "k.s"
            subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", "-o", str(out), f"-I{ROOT / 'include'}", str(f)],
                           check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            errs = check(out.read_text())
        print(f"{f.name}: {'ok' if not errs else str(len(errs)) + ' violation(s)'}")
        for e in errs[:10]:
            print("   ", e)
        rc |= bool(errs)
    return rc


if __name__ == "__main__":
    sys.exit(main())
