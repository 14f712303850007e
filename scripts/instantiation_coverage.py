#!/usr/bin/env python3
"""Which lp_step_kernel<VEC, MODE, PH, X0W, RNG, ST, ES> instantiations does the product library hold, and which of them
do the GPU parity tests really launch?

    # on the GPU box, with the coverage build of the library (python -m lanpaint_amd.build --trace):
    LANPAINT_AMD_LIB=build/liblanpaint_hip_trace.so LANPAINT_AMD_TRACE_FILE=gpurun_out/trace.txt python -m pytest tests -m gpu -q
    # anywhere:
    python scripts/instantiation_coverage.py gpurun_out/trace.txt profiles/r03_instantiation_coverage.json

The JSON is committed; tests/test_cabi_exports.py::test_every_step_kernel_instantiation_is_launched_by_a_gpu_test fails when
the library gains an instantiation the file does not list as launched (add a test that reaches it and regenerate, or drop
the instantiation)."""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "lanpaint_amd", "liblanpaint_hip.so")
FIELDS = ("VEC", "MODE", "PH", "X0W", "RNG", "ST", "ES")
MODES = {"0": "row table, fp32 mask (soft values take the general branch)", "1": "per-element times", "2": "row table, bit-packed hard mask"}
PHASES = {1: "REPLACE", 2: "POST_FIRST", 4: "POST_STEADY", 8: "PRE_HALF", 16: "EMIT", 32: "COEFFS"}


def product_instantiations(lib=LIB):
    """Template argument lists of every lp_step_kernel the host library carries a launch stub for (nm on the host symbols)."""
    out = subprocess.run(["nm", "-C", lib], check=True, capture_output=True, text=True).stdout
    found = set()
    for m in re.finditer(r"__device_stub__lp_step_kernel<([^>]*)>", out):
        found.add(", ".join(a.strip() for a in m.group(1).split(",")))
    return found


# ---- the DEVICE side: what the gfx950 code objects inside the library really contain ------------------------------------
# A host launch stub exists only for a kernel some host code names; a kernel can be instantiated on the device side without
# one (round 3: eight lp_step_kernel<..., ES = 1> code objects came from statements behind an `if constexpr` that returned).
# So the set is read from the code objects themselves: ELF section .hip_fatbin of the .so -> clang offload bundles -> the
# hipv4-amdgcn-amd-amdhsa--gfx950 entries (AMDGPU ELF) -> FUNC symbols of their .symtab.  Pure Python: no ROCm tool needed.
def _elf_sections(blob):
    """{name: (offset, size, link, entsize)} of an ELF64 little-endian image."""
    import struct
    if blob[:4] != b"\x7fELF" or blob[4] != 2 or blob[5] != 1:
        raise ValueError("not an ELF64 LE image")
    shoff, = struct.unpack_from("<Q", blob, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", blob, 0x3A)
    raw = []
    for i in range(shnum):
        name, _ty, _fl, _addr, off, size, link, _info, _al, entsize = struct.unpack_from("<IIQQQQIIQQ", blob, shoff + i * shentsize)
        raw.append((name, off, size, link, entsize))
    stroff = raw[shstrndx][1]
    out = {}
    for i, (name, off, size, link, entsize) in enumerate(raw):
        end = blob.index(b"\0", stroff + name)
        out.setdefault(blob[stroff + name:end].decode(), (off, size, link, entsize, i))
    out["__by_index__"] = raw
    return out


def _code_objects(lib, target_prefix="hipv4-amdgcn-amd-amdhsa--gfx950"):
    """The device ELF images bundled into `lib` for the target."""
    import struct
    blob = open(lib, "rb").read()
    off, size = _elf_sections(blob)[".hip_fatbin"][:2]
    fat = blob[off:off + size]
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    images, pos = [], fat.find(magic)
    while pos >= 0:
        n, = struct.unpack_from("<Q", fat, pos + len(magic))
        p = pos + len(magic) + 8
        for _ in range(n):
            e_off, e_size, t_size = struct.unpack_from("<QQQ", fat, p)
            triple = fat[p + 24:p + 24 + t_size].decode()
            p += 24 + t_size
            if triple.startswith(target_prefix) and e_size:
                images.append(fat[pos + e_off:pos + e_off + e_size])
        pos = fat.find(magic, pos + len(magic))
    return images


def _func_symbols(image):
    import struct
    sec = _elf_sections(image)
    if ".symtab" not in sec:
        return []
    off, size, link, entsize = sec[".symtab"][:4]
    stroff = sec["__by_index__"][link][1]
    names = []
    for i in range(size // (entsize or 24)):
        st_name, st_info = struct.unpack_from("<IB", image, off + i * (entsize or 24))
        if (st_info & 0xF) == 2:                                  # STT_FUNC (the kernel; its descriptor <name>.kd is an OBJECT)
            end = image.index(b"\0", stroff + st_name)
            names.append(image[stroff + st_name:end].decode())
    return names


_MANGLED = re.compile(r"^_ZN2lp14lp_step_kernelILi(\d+)ELi(\d+)ELj(\d+)ELi(\d+)ELi(\d+)ELb([01])ELi(\d+)EE")


def device_kernels(lib=LIB):
    """Mangled names of every kernel (FUNC symbol) in the gfx950 code objects of `lib`."""
    names = set()
    for image in _code_objects(lib):
        names.update(_func_symbols(image))
    return names


def kernel_resources(lib=LIB):
    """{mangled kernel name: its AMDGPU metadata record} from the NT_AMDGPU_METADATA notes (msgpack) of the gfx950 code objects:
    `.vgpr_count`, `.sgpr_count`, `.private_segment_fixed_size` (scratch bytes per lane), `.vgpr_spill_count`, ... -- what
    `hipcc -S` prints as `; ScratchSize` etc., read from the library that ships."""
    import struct
    import msgpack
    out = {}
    for image in _code_objects(lib):
        off, size = _elf_sections(image)[".note"][:2]
        p = off
        while p < off + size:
            namesz, descsz, ty = struct.unpack_from("<III", image, p)
            p += 12 + ((namesz + 3) & ~3)
            desc = image[p:p + descsz]
            p += (descsz + 3) & ~3
            if ty == 32:                                          # NT_AMDGPU_METADATA
                for k in msgpack.unpackb(desc, raw=False).get("amdhsa.kernels", []):
                    out[k[".name"]] = {a: b for a, b in k.items() if a != ".args"}
    return out


def step_kernel_args(name):
    """(VEC, MODE, PH, X0W, RNG, ST, ES) of a mangled lp_step_kernel name, or None."""
    m = _MANGLED.match(name)
    return tuple(int(g) for g in m.groups()) if m else None


def device_instantiations(lib=LIB):
    """Template argument lists of every lp_step_kernel<...> the DEVICE code of `lib` contains, in the spelling of
    product_instantiations()."""
    found = set()
    for name in device_kernels(lib):
        m = _MANGLED.match(name)
        if m:
            v, mode, ph, x0w, rng, st, es = m.groups()
            found.add(f"{v}, {mode}, {ph}u, {x0w}, {rng}, {'true' if st == '1' else 'false'}, {es}")
    return found


def describe(inst):
    a = [x.strip() for x in inst.split(",")]
    ph = int(a[2].rstrip("u"))
    return {"args": inst, "VEC": int(a[0]), "MODE": MODES.get(a[1], a[1]),
            "PH": "|".join(n for b, n in PHASES.items() if ph & b) or "run-time phases",
            "X0W": {"0": "run-time dtype", "2": "bf16/fp16 heads", "4": "fp32 heads"}[a[3]],
            "RNG": {"0": "Philox2x32", "1": "torch stream", "2": "run-time"}[a[4]], "ST": a[5] == "true",
            "ES": {"0": "off", "1": "on (decision kernel follows)", "2": "on (verdict folded into the launch)"}[a[6]]}


def main():
    trace, dst = sys.argv[1], sys.argv[2]
    launched = {ln.strip() for ln in open(trace) if ln.strip()}
    have, stubs = device_instantiations(), product_instantiations()
    doc = {"_doc": "lp_step_kernel<%s> instantiations in the gfx950 code objects of liblanpaint_hip.so (read from the bundled "
                   "device ELF images, not from the host stubs) and the ones the GPU test suite launched (coverage build of "
                   "the same sources, scripts/instantiation_coverage.py)" % ", ".join(FIELDS),
           "library_bytes": os.path.getsize(LIB), "count": len(have), "host_stub_count": len(stubs),
           "device_only": sorted(have - stubs), "host_only": sorted(stubs - have),
           "launched_count": len(have & launched),
           "instantiations": [describe(i) for i in sorted(have)],
           "launched_by_gpu_tests": sorted(have & launched), "never_launched": sorted(have - launched),
           "launched_but_not_in_library": sorted(launched - have)}
    json.dump(doc, open(dst, "w"), indent=1)
    print(f"{len(have)} device-side instantiations ({len(stubs)} host stubs, {os.path.getsize(LIB)} bytes), {len(have & launched)} launched by the GPU tests, never launched: {len(have - launched)}")
    for i in sorted(have - launched):
        print("  never launched:", i)


if __name__ == "__main__":
    main()
