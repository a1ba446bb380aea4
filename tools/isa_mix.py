#!/usr/bin/env python3
"""Static instruction mix of one gfx950 kernel, per basic block.

    python tools/isa_mix.py xattn3.hip id_xattn3_kernelILi77ELi4E
    python tools/isa_mix.py attn.hip self_attn_kernelILi40ELi1ELi4ELb0E --min-block 40

Compiles consistentid_amd/csrc/<file> with the product's flags (consistentid_amd.build.FLAGS / FILE_FLAGS) and
--save-temps into a scratch directory, cuts the named kernel out of the device assembly and counts instructions by issue
class.  No GPU needed: this is what the compiler emitted, not what a wave executed -- loops count once (the back edges are
listed so that trip counts can be applied by hand); fully unrolled kernels such as id_xattn3 read off directly.
The classes follow tools/probes/issue_rates.hip: MFMA, transcendental VALU, conversion / max / other
non-FMA VALU, FMA-class VALU, LDS, vector memory, scalar."""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

FMA_CLASS = ("v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mad", "v_fma_mix")


def classify(op: str) -> str:
    if "mfma" in op:
        return "mfma"
    if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")):
        return "valu.transcendental"
    if op.startswith("v_cvt"):
        return "valu.cvt"
    if op.startswith(("v_max", "v_min", "v_med3")):
        return "valu.minmax"
    if op.startswith(FMA_CLASS):
        return "valu.fma32"
    if op.startswith("v_pk_"):
        return "valu.packed"
    if op.startswith("v_dot"):
        return "valu.dot"
    if op.startswith(("v_mov", "v_accvgpr", "v_swap")):
        return "valu.mov"
    if op.startswith(("v_permlane", "v_readlane", "v_readfirstlane", "v_writelane", "v_perm_b32", "v_mbcnt")):
        return "valu.crosslane"
    if op.startswith(("v_cndmask", "v_cmp", "v_bfi", "v_bitop", "v_pack")):
        return "valu.select"
    if op.startswith("v_"):
        return "valu.int"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "s.wait"
    if op.startswith("s_barrier"):
        return "s.barrier"
    if op.startswith(("s_cbranch", "s_branch")):
        return "s.branch"
    if op.startswith("s_"):
        return "salu"
    return "other"


def device_asm(src: str) -> str:
    from consistentid_amd import build
    tmp = tempfile.mkdtemp(prefix="isa_mix_")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, *build.FLAGS, *build.FILE_FLAGS.get(src, []), "--save-temps=obj", "-c",
           os.path.join(ROOT, "consistentid_amd", "csrc", src), "-o", os.path.join(tmp, "k.o")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        sys.exit(r.stderr)
    for n in os.listdir(tmp):
        if n.endswith("gfx950.s"):
            with open(os.path.join(tmp, n)) as f:
                return f.read()
    sys.exit("no device assembly produced")


def kernel_body(asm: str, needle: str):
    lines = asm.split("\n")
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and needle in l]
    if len(starts) != 1:
        names = [l.split(":")[0] for l in lines if re.match(r"^_Z\S*:", l)]
        sys.exit(f"{len(starts)} kernels match {needle!r}; kernels in this file:\n  " + "\n  ".join(names))
    i = starts[0]
    j = next(k for k in range(i, len(lines)) if lines[k].strip() == "s_endpgm")
    meta = {}
    for l in lines[j:j + 400]:
        m = re.match(r"\s*\.amdhsa_next_free_vgpr\s+(\d+)", l) or re.match(r"\s*; NumVgprs:\s+(\d+)", l)
        if m:
            meta["vgprs"] = int(m.group(1))
        m = re.match(r"\s*; ScratchSize:\s+(\d+)", l)
        if m:
            meta["scratch"] = int(m.group(1))
        m = re.match(r"\s*; LDSByteSize:\s+(\d+)", l)
        if m:
            meta["lds"] = int(m.group(1))
        if "Occupancy" in l:
            meta["occupancy"] = l.split(":")[-1].strip()
            break
    return lines[i].split(":")[0], lines[i + 1:j + 1], meta


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("source")
    ap.add_argument("kernel", help="substring of the mangled kernel name")
    ap.add_argument("--min-block", type=int, default=20, help="print basic blocks with at least this many instructions")
    ap.add_argument("--asm", help="read this device assembly (hipcc --save-temps) instead of compiling csrc/<source>")
    a = ap.parse_args()
    if a.asm:
        with open(a.asm) as f:
            asm = f.read()
    else:
        asm = device_asm(a.source)
    name, body, meta = kernel_body(asm, a.kernel)
    blocks, cur = [], ["(entry)", collections.Counter(), []]
    for l in body:
        t = l.strip()
        if not t or t.startswith(";") or (t.startswith(".") and not t.endswith(":")):
            continue
        if t.endswith(":"):
            blocks.append(cur)
            cur = [t[:-1], collections.Counter(), []]
            continue
        op = t.split()[0]
        cur[1][classify(op)] += 1
        if op.startswith(("s_cbranch", "s_branch")):
            cur[2].append(t.split()[-1])
    blocks.append(cur)
    order = {b[0]: k for k, b in enumerate(blocks)}
    print(f"# {a.source} :: {name}")
    print("# " + ", ".join(f"{k}={v}" for k, v in meta.items()))
    total = collections.Counter()
    for k, (lab, c, br) in enumerate(blocks):
        total += c
        n = sum(c.values())
        if n < a.min_block:
            continue
        back = [t for t in br if t in order and order[t] <= k]
        print(f"{lab:<12} {n:5d} instr" + (f"  LOOP back edge -> {','.join(back)}" if back else ""))
        print("    " + "  ".join(f"{g}={v}" for g, v in sorted(c.items(), key=lambda kv: -kv[1])))
    n = sum(total.values())
    valu = sum(v for g, v in total.items() if g.startswith("valu."))
    print(f"TOTAL {n} instr (static)")
    print("    " + "  ".join(f"{g}={v}" for g, v in sorted(total.items(), key=lambda kv: -kv[1])))
    if total["mfma"]:
        print(f"    VALU {valu} / MFMA {total['mfma']} = {valu / total['mfma']:.2f} static")


if __name__ == "__main__":
    main()
