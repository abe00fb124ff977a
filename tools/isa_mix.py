"""Static instruction mix of the hot loops of libk22hip.so's kernels (no GPU needed).

    python tools/isa_mix.py 'conv3_halo_spec_kernelItLi256E' [more regexes ...]      > profiles/rNN_isa_mix.txt

For each kernel whose mangled name matches: the gfx950 code object is extracted from a COPY of the library (llvm-objdump --offloading),
disassembled, and every backward branch is taken as a loop [target, branch].  For the loops that contain MFMAs - innermost first - the
script prints the instruction mix and two issue-cycle bounds per iteration and per wave:

  mfma cycles   = sum over MFMAs of the matrix-pipe occupancy of the instruction (32x32x16 16-bit: 32 cycles = 8 passes; 32x32x8 / 16x16x32:
                  16; 32x32x2 f32: 64 ... - MI355X_MICROARCH.md's table; 1 wave per SIMD owns the pipe in these kernels)
  other issue   = 4 cycles per VALU instruction (a 64-wide wave over 16 lanes; packed and transcendental forms counted the same);
  lds           = LDS-array cycles of the wave's ds instructions (ds_read_b128 4, b64 / b32 2, ds_write_b32 4 ... - the guide's LDS table).
                  The array is ONE per CU: multiply by the waves of the workgroup that run this loop (4 consumer waves in the conv / GEMM
                  kernels) before comparing with the per-SIMD mfma cycles.

`mfma / (mfma + valu)` is the pipe share the loop could reach if every VALU instruction issued in the shadow of an MFMA of ANOTHER wave
(it cannot: a wave's own VALU and MFMA issue in order) - and `mfma / max(mfma, valu + salu)` the share with perfect dual issue within
the wave.  What the PMC counter measured sits below both; the gap is the barrier / DMA wait, which no static count sees.
"""
import collections
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
LIB = os.environ.get("K22_LIB_PATH", os.path.join(ROOT, "kandinsky-2_amd", "libk22hip.so"))

# matrix-pipe cycles per instruction (passes x 4), keyed by a regex on the mnemonic
MFMA_CYCLES = [
    (r"v_mfma_f32_32x32x16_(bf16|f16)", 32), (r"v_mfma_f32_16x16x32_(bf16|f16)", 16),
    (r"v_mfma_f32_32x32x8_?(bf16|f16)", 32), (r"v_mfma_f32_16x16x16_?(bf16|f16)", 16),
    (r"v_mfma_f32_32x32x2_?f32", 64), (r"v_mfma_f32_16x16x4_?f32", 32),
    (r"v_mfma_f32_32x32x4_xf32", 32), (r"v_mfma_f32_16x16x8_xf32", 16),
]
# LDS cycles per wave-instruction (MI355X_MICROARCH.md, LDS table); the port is shared by the CU's waves
LDS_READ = {"b128": 4, "b96": 8, "b64": 2, "b32": 2}
LDS_WRITE = {"b128": 13, "b96": 10, "b64": 6, "b32": 4}


def classify(mn):
    if mn.startswith("v_mfma") or mn.startswith("v_smfmac"):
        return "mfma"
    if mn.startswith("ds_read") or mn.startswith("ds_load"):
        return "ds_read"
    if mn.startswith("ds_write") or mn.startswith("ds_store"):
        return "ds_write"
    if mn.startswith("ds_"):
        return "ds_other"
    if mn.startswith(("global_load_lds", "buffer_load")) and "lds" in mn:
        return "lds_dma"
    if mn.startswith(("global_load", "buffer_load", "flat_load")):
        return "vmem_load"
    if mn.startswith(("global_store", "buffer_store", "flat_store", "global_atomic", "buffer_atomic")):
        return "vmem_store"
    if mn == "s_waitcnt":
        return "s_waitcnt"
    if mn == "s_barrier":
        return "s_barrier"
    if mn in ("s_nop", "s_sleep", "s_setprio", "s_sethalt"):
        return "s_nop"
    if mn.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if mn.startswith("s_load") or mn.startswith("s_buffer_load"):
        return "smem"
    if mn.startswith("s_"):
        return "salu"
    if mn.startswith("v_accvgpr") or mn.startswith("v_mov") or mn.startswith("v_readlane") or mn.startswith("v_readfirstlane"):
        return "valu_mov"
    if mn.startswith("v_"):
        return "valu"
    return "other"


def mfma_cycles(mn):
    for pat, c in MFMA_CYCLES:
        if re.match(pat, mn):
            return c
    return 16


def lds_cycles(mn):
    m = re.search(r"_(b128|b96|b64|b32)", mn)
    tab = LDS_WRITE if ("write" in mn or "store" in mn) else LDS_READ
    w = tab.get(m.group(1), 2) if m else 2            # sub-dword and atomic forms: counted as one 4-byte access
    if re.search(r"(read|load)2(st64)?_b64", mn):
        return 8
    return 2 * w if re.search(r"(read|write|load|store)2", mn) else w


def disassemble(so_copy, workdir):
    subprocess.run([LLVM + "/llvm-objdump", "--offloading", so_copy], check=True, capture_output=True, cwd=workdir)
    kernels = {}
    for co in sorted(glob.glob(so_copy + ".*gfx950")):
        txt = subprocess.run([LLVM + "/llvm-objdump", "-d", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout
        name, body = None, []
        for ln in txt.splitlines():
            m = re.match(r"^[0-9a-f]+ <([^>]+)>:$", ln)
            if m:
                if name and body:
                    kernels[name] = body
                name, body = m.group(1), []
                continue
            m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):(.*)$", ln)
            if m and name:    # the branch target `<kernel+0xOFF>` is printed after the encoding, in the comment
                body.append((int(m.group(3), 16), m.group(1), m.group(2) + " " + " ".join(re.findall(r"<[^>]+>", m.group(4)))))
        if name and body:
            kernels[name] = body
    return kernels


def loops_of(body):
    """[(start_index, end_index)] of every backward branch, innermost (shortest) first; branch targets are printed as `<kernel+0xOFF>`."""
    addr_index = {a: i for i, (a, _, _) in enumerate(body)}
    base = body[0][0]
    out = []
    for i, (a, mn, ops) in enumerate(body):
        if not mn.startswith(("s_cbranch", "s_branch")):
            continue
        m = re.search(r"\+0x([0-9a-fA-F]+)>", ops)
        tgt = base + int(m.group(1), 16) if m else (base if re.search(r"<[^+>]+>", ops) else None)
        if tgt is not None and tgt in addr_index and tgt <= a:
            out.append((addr_index[tgt], i))
    return sorted(set(out), key=lambda se: se[1] - se[0])


def mix(body, s, e):
    c, cyc = collections.Counter(), collections.Counter()
    kinds = collections.Counter()
    for _, mn, _ in body[s:e + 1]:
        k = classify(mn)
        c[k] += 1
        if k == "mfma":
            cyc["mfma"] += mfma_cycles(mn)
            kinds[mn] += 1
        elif k in ("valu", "valu_mov"):
            cyc["valu"] += 4
        elif k in ("ds_read", "ds_write", "ds_other"):
            cyc["lds"] += lds_cycles(mn)
            kinds[mn] += 1
        elif k in ("salu", "branch", "s_nop", "s_waitcnt", "s_barrier", "smem"):
            cyc["salu"] += 1
        elif k in ("lds_dma", "vmem_load", "vmem_store"):
            cyc["vmem"] += 4
            kinds[mn] += 1
    return c, cyc, kinds


def report(name, body, out):
    loops = [(s, e) for s, e in loops_of(body) if any(classify(mn) == "mfma" for _, mn, _ in body[s:e + 1])]
    tot, _, _ = mix(body, 0, len(body) - 1)
    print(f"\n{name}\n  {len(body)} instructions, {tot['mfma']} MFMAs in the whole kernel, {len(loops)} loop(s) with MFMAs", file=out)
    seen = []
    for s, e in loops:
        nested = [l for l in seen if s <= l[0] and l[1] <= e]
        seen.append((s, e))
        c, cyc, kinds = mix(body, s, e)
        m, v, sa, l = cyc["mfma"], cyc["valu"], cyc["salu"], cyc["lds"]
        print(f"  loop @{body[s][0]:#x}..{body[e][0]:#x}: {e - s + 1} instr" + (f" (contains {len(nested)} inner loop(s))" if nested else ""), file=out)
        print("    mix: " + ", ".join(f"{k} {c[k]}" for k in ("mfma", "valu", "valu_mov", "ds_read", "ds_write", "lds_dma", "vmem_load", "vmem_store",
                                                                  "salu", "s_waitcnt", "s_barrier", "s_nop", "branch") if c[k]), file=out)
        print("    ops: " + ", ".join(f"{k} x{n}" for k, n in sorted(kinds.items(), key=lambda kv: -kv[1])), file=out)
        if m:
            print(f"    per wave and iteration: mfma {m} cycles, valu {v}, salu/waits {sa}, lds array {l} (x4 waves: {4 * l / m:.2f} of the mfma cycles), vmem issue {cyc['vmem']}  ->  "
                  f"mfma/(mfma+valu) = {m / (m + v):.2f}, mfma/max(mfma, valu+salu) = {m / max(m, v + sa):.2f}", file=out)


def main(patterns):
    with tempfile.TemporaryDirectory() as td:
        so = os.path.join(td, "lib.so")
        shutil.copy(LIB, so)
        kernels = disassemble(so, td)
    print(f"# static instruction mix, {os.path.relpath(LIB, ROOT)} ({len(kernels)} gfx950 functions); tools/isa_mix.py {' '.join(patterns)}")
    for pat in patterns:
        hit = [k for k in sorted(kernels) if re.search(pat, k)]
        print(f"\n## /{pat}/: {len(hit)} kernel(s)")
        for k in hit:
            report(k, kernels[k], sys.stdout)


if __name__ == "__main__":
    main(sys.argv[1:] or ["conv3_halo_spec_kernelItLi256E"])
