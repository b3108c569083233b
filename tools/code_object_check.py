"""Read the gfx950 code objects inside libga_hip.so and report, per kernel, the numbers the compiler recorded in the
code-object metadata: private segment (scratch) bytes, VGPR / SGPR spill counts, VGPR / AGPR / SGPR use, LDS bytes.

    python tools/code_object_check.py [path/to/libga_hip.so]       # table + the kernels that use scratch or spill

Needs only the LLVM tools of the ROCm image (llvm-objcopy, llvm-readelf); no GPU.  tests/test_abi.py asserts through
`kernels()` that no kernel of the library uses scratch memory or spills registers."""
import re
import struct
import subprocess
import sys
import tempfile
from pathlib import Path

LLVM = Path("/opt/rocm/lib/llvm/bin")
BUNDLE_MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
FIELDS = ("private_segment_fixed_size", "vgpr_spill_count", "sgpr_spill_count", "vgpr_count", "agpr_count", "sgpr_count",
          "group_segment_fixed_size")


def _device_elfs(fatbin):
    """Every gfx9xx ELF of the (possibly concatenated) clang offload bundles in the .hip_fatbin section."""
    out, pos = [], 0
    while True:
        pos = fatbin.find(BUNDLE_MAGIC, pos)
        if pos < 0:
            return out
        (n,) = struct.unpack_from("<Q", fatbin, pos + len(BUNDLE_MAGIC))
        cur = pos + len(BUNDLE_MAGIC) + 8
        for _ in range(n):
            off, size, idlen = struct.unpack_from("<QQQ", fatbin, cur)
            ident = fatbin[cur + 24:cur + 24 + idlen].decode()
            cur += 24 + idlen
            if "amdgcn" in ident and size:
                out.append((ident, fatbin[pos + off:pos + off + size]))
        pos = cur


def extract_code_objects(lib_path, out_dir):
    """Writes every device ELF of the library to out_dir/co<i>.elf and returns the paths (for llvm-objdump and friends)."""
    fat = Path(out_dir) / "fat.bin"
    subprocess.run([str(LLVM / "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", str(lib_path), str(fat)], check=True)
    paths = []
    for i, (_ident, elf) in enumerate(_device_elfs(fat.read_bytes())):
        p = Path(out_dir) / f"co{i}.elf"
        p.write_bytes(elf)
        paths.append(p)
    return paths


def kernels(lib_path):
    """[{name, private_segment_fixed_size, vgpr_spill_count, ...}] for every kernel of every device code object."""
    lib_path = Path(lib_path)
    found = []
    with tempfile.TemporaryDirectory() as tmp:
        fat = Path(tmp) / "fat.bin"
        subprocess.run([str(LLVM / "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", str(lib_path), str(fat)],
                       check=True)
        for i, (_ident, elf) in enumerate(_device_elfs(fat.read_bytes())):
            p = Path(tmp) / f"co{i}.elf"
            p.write_bytes(elf)
            notes = subprocess.run([str(LLVM / "llvm-readelf"), "--notes", str(p)], check=True, capture_output=True,
                                   text=True).stdout
            cur = None
            for line in notes.splitlines():
                m = re.match(r"(  - |    )\.(\w+):\s*(.*)$", line)     # kernel-level keys only (arguments sit deeper)
                if not m:
                    continue
                key, val = m.group(2), m.group(3).strip().strip("'")
                if m.group(1) == "  - ":                                  # first key of a kernel entry
                    cur = {"object": i}
                    found.append(cur)
                if cur is None:
                    continue
                if key in FIELDS:
                    cur[key] = int(val)
                elif key == "name":
                    cur["name"] = val
                elif key == "symbol":
                    cur["symbol"] = val
    return [k for k in found if "name" in k]


def offenders(ks):
    """Kernels that touch scratch memory: a private segment (stack arrays the compiler could not keep in registers) or
    spilled vector registers.  Spilled SGPRs are parked in VGPR lanes (no memory traffic) and are only reported."""
    return [k for k in ks if k.get("private_segment_fixed_size", 0) or k.get("vgpr_spill_count", 0)]


def main():
    args = [a for a in sys.argv[1:] if a != "-v"]
    lib = args[0] if args else Path(__file__).resolve().parents[1] / "guided-attention_amd" / "libga_hip.so"
    ks = kernels(lib)
    bad = offenders(ks)
    sg = [k for k in ks if k.get("sgpr_spill_count", 0) and k not in bad]
    print(f"{len(ks)} kernels in {lib}; {len(bad)} use scratch memory or spill VGPRs; {len(sg)} park SGPRs in VGPR lanes")
    for k in bad + (sg if "-v" in sys.argv else []):
        demangled = subprocess.run(["c++filt", k["name"]], capture_output=True, text=True).stdout.strip() or k["name"]
        print(f"  scratch {k.get('private_segment_fixed_size', 0):5d} B  vgpr spills {k.get('vgpr_spill_count', 0):3d}  "
              f"sgpr spills {k.get('sgpr_spill_count', 0):3d}  vgprs {k.get('vgpr_count', 0):3d}  {demangled[:150]}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
