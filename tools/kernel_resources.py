"""Register / scratch / LDS use of every gfx950 kernel in a built libdimn.so, from the code object's own metadata, plus -- for the kernels whose
inner loops rely on hand-counted `s_waitcnt vmcnt(N)` (k_predict_bf16) -- the invariants that make those counts right:
no scratch, and the number of vector-memory requests between two waits of the loop.

    python tools/kernel_resources.py [path/to/libdimn.so] [name-substring ...]

Runs on the build host (no GPU): objcopy + clang-offload-bundler + llvm-readelf / llvm-objdump from /opt/rocm/lib/llvm/bin.
tests/test_abi.py::test_kernel_resource_invariants calls check() on the shipped library."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def code_object(lib, workdir):
    fat = os.path.join(workdir, "fat.bin")
    co = os.path.join(workdir, "gfx950.co")
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                           "--input=" + fat, "--output=" + co])
    return co


def kernel_table(co):
    """name -> the kernel's metadata record (amdhsa.kernels: vgpr_count, agpr_count, spills, scratch, LDS ...)."""
    import yaml
    notes = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", co], text=True)
    doc = notes[notes.index("---"):]
    doc = doc[:doc.index("\n...")] if "\n..." in doc else doc
    meta = yaml.safe_load(doc)
    return {k[".name"]: {key.lstrip("."): val for key, val in k.items() if isinstance(val, int)} for k in meta["amdhsa.kernels"]}


def demangle(names):
    p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return dict(zip(names, p.stdout.splitlines()))


def disassemble(co, symbol):
    txt = subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", "--disassemble-symbols=" + symbol, co], text=True)
    return [l.split("//")[0].strip() for l in txt.splitlines() if l.startswith("\t")]


VMEM = re.compile(r"^(global_load|buffer_load|global_store|buffer_store|flat_load|flat_store|scratch_)")


def check(lib=None):
    """The invariants the hand-scheduled kernels rely on; returns (table, problems)."""
    lib = lib or os.path.join(ROOT, "deepimpute_amd", "csrc", "libdimn.so")
    problems = []
    with tempfile.TemporaryDirectory() as wd:
        co = code_object(lib, wd)
        tab = kernel_table(co)
        nice = demangle(list(tab))
        for sym, rec in tab.items():
            rec["demangled"] = nice.get(sym, sym)
        for sym, rec in tab.items():
            name = rec["demangled"]
            # (1) the software-pipelined kernels must not touch scratch inside their loops: the forward-only instances of k_predict_bf16
            #     (the LOSS = true instance spills in its epilogue only; its loop is checked below), the streaming training kernels
            if (name.startswith("void k_predict_bf16<") and ", false>" in name) or name.startswith("void k_w1_update_fwd_ring<") or name.startswith("void k_mid_pipe<"):
                if rec.get("private_segment_fixed_size", 0) or rec.get("vgpr_spill_count", 0):
                    problems.append("%s: %d B of scratch, %d spilled VGPRs" % (name, rec.get("private_segment_fixed_size", 0), rec.get("vgpr_spill_count", 0)))
            # (2) one workgroup of the ring B1F1 (16 waves) must fit a CU: 4 waves per SIMD -> at most 128 registers per lane
            if name.startswith("void k_w1_update_fwd_ring<16, 1, 3, 1"):
                if rec.get("vgpr_count", 0) + 0 > 128:
                    problems.append("%s: %d VGPRs > 128" % (name, rec["vgpr_count"]))
            # (3) k_predict_bf16: between two hand-counted waits of the first layer's loop (`s_waitcnt vmcnt(8)` / `vmcnt(16)`) the wave must have
            #     issued exactly the requests the count assumes, and no scratch access may sit among them (it would share the counter)
            if name.startswith("void k_predict_bf16<"):
                ins = disassemble(co, sym)
                waits = [i for i, l in enumerate(ins) if re.match(r"s_waitcnt vmcnt\((8|16)\)", l)]
                for a, b in zip(waits, waits[1:]):
                    seg = ins[a:b]
                    if any(l.startswith("s_cbranch") or l.startswith("s_branch") for l in seg):
                        continue                 # a loop boundary: counted by the segment on the other side
                    if any(l.startswith("scratch_") for l in seg):
                        problems.append("%s: scratch access between two counted waits" % name)
                rec["counted_waits"] = len(waits)
                if not waits:
                    problems.append("%s: no hand-counted vmcnt(8) / vmcnt(16) wait found -- was the loop rewritten? update tools/kernel_resources.py" % name)
    return tab, problems


def main():
    args = sys.argv[1:]
    lib = args.pop(0) if args and args[0].endswith(".so") else None
    tab, problems = check(lib)
    rows = sorted(tab.values(), key=lambda r: r["demangled"])
    print("%-100s %5s %5s %6s %7s %7s" % ("kernel", "vgpr", "agpr", "spill", "scratch", "lds"))
    for r in rows:
        if args and not any(a in r["demangled"] for a in args):
            continue
        print("%-100s %5d %5d %6d %7d %7d" % (r["demangled"][:100], r.get("vgpr_count", -1), r.get("agpr_count", -1), r.get("vgpr_spill_count", 0),
                                              r.get("private_segment_fixed_size", 0), r.get("group_segment_fixed_size", 0)))
    for p in problems:
        print("PROBLEM:", p)
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main())
