#!/usr/bin/env python3
"""Static resource table of every gfx950 kernel in libhabitat_amd.so: VGPRs, AGPRs, SGPRs, static LDS, scratch bytes, spilled VGPRs and the
workgroup-size bound, read from the code objects' amdhsa metadata (no GPU needed).

  python tools/kernel_resources.py [path/to/libhabitat_amd.so] > profiles/rNN_kernel_resources.txt

How: the `.hip_fatbin` section holds one clang offload bundle per translation unit; each is unbundled with clang-offload-bundler and its
notes are read with llvm-readelf.  `vgpr` is the unified count (architectural registers + the accumulation registers listed under `agpr`);
`waves/SIMD` is the register-file bound alone (512 unified registers per lane and SIMD, allocation granule 8, at most 8 waves) -- dynamic LDS (most strip kernels ask for it at launch) and the workgroup size lower it further.
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "habitat-lab_amd", "habitat_amd", "libhabitat_amd.so")
    rows = []
    with tempfile.TemporaryDirectory() as d:
        fat = os.path.join(d, "fatbin")
        subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
        data = open(fat, "rb").read()
        offs = [m.start() for m in re.finditer(rb"__CLANG_OFFLOAD_BUNDLE__", data)] + [len(data)]
        for i in range(len(offs) - 1):
            part, co = os.path.join(d, f"fb{i}"), os.path.join(d, f"co{i}.o")
            open(part, "wb").write(data[offs[i]:offs[i + 1]])
            subprocess.run([f"{LLVM}/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={part}",
                            f"--output={co}", "--unbundle"], check=True)
            notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
            for blk in notes.split("  - .agpr_count:")[1:]:
                blk = "  - .agpr_count:" + blk
                g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "0"])[1]
                rows.append([g("name")] + [int(g(k)) for k in ("vgpr_count", "agpr_count", "sgpr_count", "group_segment_fixed_size",
                                                                "private_segment_fixed_size", "vgpr_spill_count", "max_flat_workgroup_size")])
    names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
    for r, n in zip(rows, names):
        r[0] = re.sub(r"\(.*$", "", n.replace("void ", "").replace("hab::", ""))
    rows.sort(key=lambda r: (-r[1], r[0]))
    print(f"# {os.path.basename(lib)}: {len(rows)} kernels; {sum(1 for r in rows if r[5] or r[6])} with scratch or spilled VGPRs")
    print(f"# {'kernel':88s} {'vgpr':>4s} {'agpr':>4s} {'sgpr':>4s} {'lds(static)':>11s} {'scratch':>7s} {'spilled':>7s} {'wg<=':>5s} {'waves/SIMD(regs)':>16s}")
    for n, v, a, s, lds, scr, sp, wg in rows:
        regs = (v + 7) // 8 * 8
        print(f"{n[:90]:90s} {v:4d} {a:4d} {s:4d} {lds:11d} {scr:7d} {sp:7d} {wg:5d} {min(8, 512 // max(regs, 1)):16d}")


if __name__ == "__main__":
    main()
