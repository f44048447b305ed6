#!/usr/bin/env python3
"""Executed instruction mix of a trace-kernel instantiation, per region of the kernel.

  static mix   : the PRODUCT build's assembly (hipcc -S -gline-tables-only with build.py's flags; line tables do not
                 change code generation), every instruction assigned to the SRT_REGION(...) stretch of csrc/kernels.hip
                 it was written in (its .loc line; inlined helpers inherit the region of the code around them) and to a
                 class (arith / int / cmp+cndmask / mov / lane / trans / fp64 / cvt / s_nop / salu / smem / lds / vmem / branch / wait).
  frequencies  : how often a wave ran each region, from a -DSRT_REGION_COUNT build on the GPU (scripts/region_probe.py).
  executed mix : static x frequency, summed per phase; checked against rocprofv3 PMC totals of the product build when given.

usage:  isa_phase_mix.py static  [--kernel 'srt_trace_kernel<false, true, false, false>'] [--flags ...]        (CPU box)
        isa_phase_mix.py merge --freq gpurun_out/region_freq.json [--pmc profiles/xyz.json] --out profiles/r03_phase_mix.json
"""
import argparse
import collections
import json
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "simple-raytracer_amd" / "csrc"
SRC = CSRC / "kernels.hip"
BASE_FLAGS = ["-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fno-gpu-rdc", "--offload-arch=gfx950"]

PHASE_OF = {  # region -> phase of DESIGN.md's split
    "PROLOGUE": "other", "EPILOGUE": "other", "LOOP_HEAD": "loop", "LOOP_TAIL": "loop",
    "EXTEND_SETUP": "extend", "EXTEND_GROUP": "extend",
    **{f"EXTEND_{n}_{k}": "extend" for n in ("SPHERES2", "SPHERES4", "PLANES", "MODEL") for k in range(3)},
    "EXTEND_SUSPEND": "extend", "EXTEND_TRI_LOOP": "extend", "EXTEND_TRI_EXACT": "extend", "EXTEND_TRI_DIV": "extend", "EXTEND_BVH_STEP": "extend", "EXTEND_FINISH": "extend",
    "SKY_PUSH": "sky", "SKY_RESOLVE": "sky",
    "SHADE_HEAD": "shade", "SHADE_POP": "shade", "SHADE_WINNER": "shade", "SHADE_MESH_NORMAL": "shade", "SHADE_MATERIAL": "shade",
    "SHADE_BOUNCE": "shade", "SHADE_OPAQUE": "shade", "SHADE_GLASS": "shade", "SHADE_REFRACT": "shade", "SHADE_TAIL": "shade", "PARK": "shade",
    "HANDIN": "handin", "HANDIN_ORPHAN": "handin",
    "REFILL_HEAD": "refill", "REFILL_SCANQ": "refill", "REFILL_LOOP": "refill", "REFILL_OPEN": "refill", "REFILL_FLUSH": "refill",
    "REFILL_CURSOR": "refill", "REFILL_TAKE": "refill", "CAMERA": "refill",
    "RARE": "rare",
}
# helper functions whose code belongs to one region wherever it is inlined (name -> region); all other helpers inherit
FIXED_HELPERS = {"sky_box": "SKY_RESOLVE", "sample_sky": "SKY_RESOLVE", "dm_atan2pif": "SKY_RESOLVE", "dm_powi": "SKY_RESOLVE",
                 "dm_powf": "RARE", "dm_log_pos_d": "RARE", "dm_exp_d": "RARE",  # general pow: only for a non-integer sun_focus
                 "moller_trumbore": None, "flush_stage": None}
CLASSES = ["arith", "int", "cmp_cndmask", "mov", "lane", "trans", "fp64", "cvt", "s_nop", "salu", "smem", "lds", "vmem", "branch", "wait"]
VALU_CLASSES = ["arith", "int", "cmp_cndmask", "mov", "lane", "trans", "fp64", "cvt"]


def classify(mn):
    if mn == "s_nop":
        return "s_nop"
    if mn.startswith("s_waitcnt") or mn in ("s_barrier", "s_sleep"):
        return "wait"
    if mn.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_swappc", "s_getpc")):
        return "branch"
    if mn.startswith(("s_load", "s_buffer_load", "s_store", "s_memtime", "s_memrealtime", "s_dcache", "s_atomic")):
        return "smem"
    if mn.startswith("s_"):
        return "salu"
    if mn.startswith("ds_"):
        return "lds"
    if mn.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if not mn.startswith("v_"):
        return "salu"
    if mn.startswith(("v_readlane", "v_writelane", "v_readfirstlane", "v_permlane", "v_mov_b32_dpp", "v_bpermute")):
        return "lane"
    if "_f64" in mn and not mn.startswith("v_cvt"):
        return "fp64"
    if mn.startswith(("v_cmp", "v_cmpx", "v_cndmask")):
        return "cmp_cndmask"
    if mn.startswith(("v_mov", "v_accvgpr", "v_swap")):
        return "mov"
    if mn.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_log", "v_exp", "v_sin", "v_cos")):
        return "trans"
    if mn.startswith("v_cvt"):
        return "cvt"
    if re.search(r"_f32|_f16|_legacy", mn) and not mn.startswith(("v_frexp", "v_ldexp")) or mn.startswith(("v_div_", "v_fma", "v_mad_f", "v_mac_f", "v_med3_f", "v_min3_f", "v_max3_f")):
        return "arith"
    return "int"


# Issue cost of a VALU instruction relative to a full-rate one (v_add_f32 / v_mul_f32 / v_xor / v_mov ... = 1), from the saturated
# throughput measured on an MI355X with 8 waves per SIMD (scripts/microbench/valu_rates2.hip, profiles/r03_valu_rates_saturated.txt):
# compares, selects, conversions, min / max / med3, left shifts, bit-field ops, three-operand integer ops, integer multiplies and
# fp64 run at 0.57 of that rate, the transcendental unit at 0.29.
FULL_RATE = ("v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_fmaak_f32", "v_fmamk_f32", "v_mac_f32", "v_xor_b32", "v_and_b32",
             "v_or_b32", "v_not_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_mov_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_accvgpr")


def valu_cost(mn):
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", mn)
    if base.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_log", "v_exp", "v_sin", "v_cos")):
        return 3.43
    if base.startswith("v_mad_u64") or base.startswith("v_mad_i64"):
        return 3.5
    if base in FULL_RATE:
        return 1.0
    return 1.75


# Round 4: what an instruction of each kind costs the LAUNCH, measured in place -- the sphere kernel at full size with 100 extra
# instructions of one kind per loop iteration (scripts/r04_issue_cost.sh, profiles/r04_issue_cost.json), in units of one
# v_add_f32. Scalar instructions are not free (the kernel's time follows the waves' whole instruction stream, not the VALU
# count alone); kinds not measured carry the figure of the nearest measured one.
def issue_cost(mn, cls):
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", mn)
    if cls == "s_nop":
        return 0.24
    if cls == "salu":
        return 1.03 if base.startswith(("s_mov", "s_movk")) else 1.32
    if cls == "branch":
        return 0.75  # s_cmp + branch: 1.4 not taken, 2.4 taken (to the next instruction); the compare is counted as salu
    if cls == "wait":
        return 0.24  # as s_nop; what it waits for is not modelled
    if cls in ("smem", "lds", "vmem"):
        return 1.3  # not measured: taken as a scalar instruction
    if cls == "lane":
        return 2.55
    if cls in ("trans", "fp64"):
        return 5.2
    if base.startswith(("v_mul_lo", "v_mul_hi", "v_mad_u32", "v_mad_i32")):
        return 1.78
    if base.startswith(("v_mad_u64", "v_mad_i64")):
        return 5.2
    if base in ("v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_fmaak_f32", "v_fmamk_f32", "v_mac_f32", "v_mov_b32"):
        return 1.0 if base != "v_fma_f32" else 1.08
    if base in ("v_xor_b32", "v_and_b32", "v_or_b32", "v_not_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_add_u32", "v_sub_u32", "v_subrev_u32"):
        return 1.31
    return 1.65  # compares, selects, conversions, min / max / med3, left shifts, bit-field and three-operand integer ops


LLVM_BIN = Path("/opt/rocm/lib/llvm/bin")


def compile_code_object(flags, out):
    """the device code object with line tables + inlining records (-gline-tables-only: same code as the product build)"""
    obj = Path(str(out) + ".o")
    cmd = ["/opt/rocm/bin/hipcc", *BASE_FLAGS, *flags, "--cuda-device-only", "-c", "-gline-tables-only", "-o", str(obj), str(SRC)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise SystemExit("hipcc failed:\n" + r.stderr)
    r = subprocess.run([str(LLVM_BIN / "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={obj}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={out}"],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise SystemExit("clang-offload-bundler failed:\n" + r.stderr)


slot_calls = {}  # source line of a test_block(..., slot) call -> slot


def source_tables():
    """(markers [(line, region)], body (lo, hi) of the trace kernel, helper ranges [(lo, hi, name)], rare lines) of kernels.hip"""
    lines = SRC.read_text().split("\n")
    markers, rare = [], set()
    slot_calls.clear()
    for i, l in enumerate(lines, 1):
        m = re.search(r"\bSRT_REGION\((\w+)\)", l) or re.search(r"\bSRT_REGION_SLOT\((\w+), slot\)", l)
        if m and not l.lstrip().startswith(("#define", "//", "X(")) and "R_##" not in l:
            markers.append((i, m.group(1) + ("_@" if "SRT_REGION_SLOT" in l else "")))
        m = re.search(r"\btest_block\(.*, (\d)\);", l)
        if m:
            slot_calls[i] = int(m.group(1))
        if "@rare" in l:
            rare.add(i)
    body_lo = next(i for i, l in enumerate(lines, 1) if "void srt_trace_kernel(const TraceParams p)" in l)
    body_hi = next(i for i, l in enumerate(lines, 1) if i > body_lo and l.startswith("}"))
    helpers = []
    fn = re.compile(r"^(?:template.*>\s*)?__device__ __forceinline__ [\w:<> ]+?[ &*](\w+)\(")
    cur = None
    for i, l in enumerate(lines, 1):
        m = fn.match(l)
        if m and i < body_lo:
            if cur:
                helpers.append((cur[0], i - 1, cur[1]))
            cur = (i, m.group(1))
    if cur:
        helpers.append((cur[0], body_lo - 1, cur[1]))
    return markers, (body_lo, body_hi), helpers, rare


def disassemble(code_object, kernel_demangled):
    """[(address, mnemonic)] of one kernel"""
    txt = subprocess.run([str(LLVM_BIN / "llvm-objdump"), "-d", "--no-show-raw-insn", str(code_object)], capture_output=True, text=True).stdout.split("\n")
    syms = [re.match(r"^[0-9a-f]+ <(\w+)>:", l).group(1) for l in txt if re.match(r"^[0-9a-f]+ <\w+>:", l)]
    dem = subprocess.run(["c++filt"], input="\n".join(syms), capture_output=True, text=True).stdout.split("\n")
    want = next((s for s, d in zip(syms, dem) if d.replace("void ", "").split("(")[0] == kernel_demangled), None)
    if want is None:
        raise SystemExit(f"kernel {kernel_demangled!r} not found; have: " + "; ".join(d for d in dem if "trace" in d))
    out, on, base = [], False, None
    for l in txt:
        if re.match(r"^[0-9a-f]+ <", l):
            on = f"<{want}>:" in l
            continue
        if on:
            m = re.match(r"^\s+(\S+).*//\s*([0-9A-F]+):", l)
            if m:
                addr = int(m.group(2), 16)
                if base is None:
                    base = addr
                t = re.search(r"<\w+\+0x([0-9a-f]+)>\s*$", l) if m.group(1).startswith(("s_cbranch", "s_branch")) else None
                tgt = (base + int(t.group(1), 16)) if t else (base if m.group(1).startswith(("s_cbranch", "s_branch")) and l.rstrip().endswith(f"<{want}>") else None)
                out.append((addr, m.group(1), l.split("//")[0].strip(), tgt))
    return out


def inline_stacks(code_object, addrs):
    """per address: [(function, file, line)] innermost first"""
    r = subprocess.run([str(LLVM_BIN / "llvm-symbolizer"), f"--obj={code_object}", "--inlines", "--output-style=LLVM"],
                       input="\n".join(hex(a) for a in addrs), capture_output=True, text=True)
    stacks, cur, fn = [], [], None
    for l in r.stdout.split("\n"):
        if not l.strip():
            if cur:
                stacks.append(cur)
            cur, fn = [], None
            continue
        if fn is None:
            fn = l.strip()
        else:
            m = re.match(r"^(.*):(\d+):(\d+)$", l.strip())
            cur.append((fn, Path(m.group(1)).name if m else "", int(m.group(2)) if m else 0))
            fn = None
    if cur:
        stacks.append(cur)
    assert len(stacks) == len(addrs), (len(stacks), len(addrs))
    return stacks


def static_mix(kernel, flags, cache="/tmp/srt_phase_mix.hsaco"):
    compile_code_object(flags, cache)
    markers, (body_lo, body_hi), helpers, rare = source_tables()
    insts = disassemble(cache, kernel)
    stacks = inline_stacks(cache, [a[0] for a in insts])

    def helper_of(line):
        for lo, hi, name in helpers:
            if lo <= line <= hi:
                return lo, hi, name
        return None

    def resolve(stack):
        """(region, copy): copy = the kernel-body call sites outside the frame that decides the region -- a helper that carries a
        marker (or a lambda) is compiled once per place it is inlined at, and the counter of its region sums over those copies"""
        region, slot, depth = None, 0, None
        frames = list(reversed(stack))  # outermost frame first: inner frames refine
        for d, (fn, file, line) in enumerate(frames):
            base = fn.split("<")[0].replace("void ", "")
            if base in FIXED_HELPERS and FIXED_HELPERS[base]:
                region, depth = FIXED_HELPERS[base], d
                continue
            if file != "kernels.hip" or line == 0:
                continue
            if line in slot_calls:
                slot = slot_calls[line]
            if line in rare:
                return "RARE", ()
            if body_lo <= line <= body_hi:
                if region is None:
                    region, depth = "PROLOGUE", d  # the kernel's lines before its first marker
                for ml, name in markers:
                    if body_lo <= ml <= line:
                        region, depth = name, d
            else:
                h = helper_of(line)
                if h:
                    for ml, name in markers:
                        if h[0] <= ml <= line:
                            region, depth = name, d
        if not region:
            return None, ()
        copy = tuple(line for fn, file, line in frames[:depth] if file == "kernels.hip" and body_lo <= line <= body_hi)
        return region.replace("_@", f"_{slot}"), copy

    # ---- basic blocks: an instruction the compiler merged from several places carries no line; it takes the region of its
    # block (majority of the block's resolved instructions), a block without any takes the region of a neighbour in the
    # control-flow graph (the one that runs most often when frequencies are known)
    starts = {insts[0][0]}
    for i, (addr, mn, text, tgt) in enumerate(insts):
        if tgt is not None:
            starts.add(tgt)
        if mn.startswith(("s_cbranch", "s_branch", "s_endpgm")) and i + 1 < len(insts):
            starts.add(insts[i + 1][0])
    res2 = [resolve(st) for st in stacks]
    res = [r for r, _ in res2]
    # what the compiler hoisted out of the main loop (constants, invariant addresses) keeps the line it came from but runs
    # once per wave: everything laid out before the loop's first instruction is prologue
    first_loop = next((i for i, r in enumerate(res) if r == "LOOP_HEAD"), 0)
    res = [("PROLOGUE" if i < first_loop and r != "RARE" else r) for i, r in enumerate(res)]
    res2 = [((r, c[1]) if r == c[0] else (r, ())) for r, c in zip(res, res2)]
    copies = collections.defaultdict(set)
    for r, c in res2:
        if r and r != "RARE":
            copies[r].add(c)
    static_mix.copies = {r: len(c) for r, c in copies.items()}
    blocks, cur = [], []
    for i, ins in enumerate(insts):
        if ins[0] in starts and cur:
            blocks.append(cur)
            cur = []
        cur.append(i)
    blocks.append(cur)
    first_of = {insts[b[0]][0]: k for k, b in enumerate(blocks)}
    succ = collections.defaultdict(set)
    for k, b in enumerate(blocks):
        addr, mn, text, tgt = insts[b[-1]]
        if tgt is not None and tgt in first_of:
            succ[k].add(first_of[tgt])
        if not mn.startswith(("s_branch", "s_endpgm")) and k + 1 < len(blocks):
            succ[k].add(k + 1)
    pred = collections.defaultdict(set)
    for k, ss in succ.items():
        for j in ss:
            pred[j].add(k)
    freq = static_mix.freq or {}
    bb_region = {}
    for k, b in enumerate(blocks):
        votes = collections.Counter(res[i] for i in b if res[i] and res[i] != "RARE")
        if votes:
            bb_region[k] = votes.most_common(1)[0][0]
        elif any(res[i] == "RARE" for i in b):
            bb_region[k] = "RARE"
    for _ in range(64):  # propagate to blocks without a line of their own
        changed = False
        for k in range(len(blocks)):
            if k in bb_region:
                continue
            cand = [bb_region[j] for j in list(succ[k]) + list(pred[k]) if j in bb_region and bb_region[j] != "RARE"]
            if cand:
                bb_region[k] = max(cand, key=lambda r: freq.get(r, 0))
                changed = True
        if not changed:
            break
    per_region = collections.defaultdict(lambda: collections.Counter())
    listing = []
    for k, b in enumerate(blocks):
        cur = None
        for i in b:
            addr, mn, text, tgt = insts[i]
            if res[i]:
                cur = res[i]
            r = cur or bb_region.get(k, "PROLOGUE")
            per_region[r][classify(mn)] += 1
            if classify(mn) in VALU_CLASSES:
                per_region[r]["valu_cost"] += valu_cost(mn)
            per_region[r]["issue_cost"] += issue_cost(mn, classify(mn))
            listing.append((addr, r, classify(mn), text, " < ".join(f"{fn.split('(')[0]}:{line}" for fn, _, line in stacks[i]), k))
    static_mix.listing = listing
    return {r: dict(c) for r, c in per_region.items()}


static_mix.freq = None
static_mix.copies = {}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("mode", choices=["static", "merge", "dump"])
    ap.add_argument("--region", default="", help="dump: regions to list (prefix match), comma separated")
    ap.add_argument("--kernel", default="srt_trace_kernel<false, true, false, false>")
    ap.add_argument("--flags", default="")
    ap.add_argument("--freq")
    ap.add_argument("--pmc")
    ap.add_argument("--out")
    a = ap.parse_args()
    if a.freq:
        static_mix.freq = {k: v[0] for k, v in json.loads(Path(a.freq).read_text())["regions"].items()}
    st = static_mix(a.kernel, a.flags.split())
    if a.mode == "dump":
        want = tuple(a.region.split(","))
        for addr, r, cls, text, stack, bb in static_mix.listing:
            if r.startswith(want):
                print(f"{addr:6x} bb{bb:<4d} {r:18s} {cls:11s} {text:60s} {stack}")
        return
    if a.mode == "static":
        tot = collections.Counter()
        print(f"{'region':18s} " + " ".join(f"{c[:6]:>6s}" for c in CLASSES) + "   VALU")
        for r in sorted(st, key=lambda r: list(PHASE_OF).index(r) if r in PHASE_OF else 99):
            c = st[r]
            tot.update(c)
            print(f"{r:18s} " + " ".join(f"{c.get(k, 0):6d}" for k in CLASSES) + f" {sum(c.get(k, 0) for k in VALU_CLASSES):6d}  x{static_mix.copies.get(r, 1)}")
        print(f"{'total':18s} " + " ".join(f"{tot.get(k, 0):6d}" for k in CLASSES) + f" {sum(tot.get(k, 0) for k in VALU_CLASSES):6d}")
        return
    fq = json.loads(Path(a.freq).read_text())
    waves = {k: v[0] for k, v in fq["regions"].items()}
    lanes = {k: v[1] for k, v in fq["regions"].items()}
    alias = {"RARE": 0}
    out_regions, phases = {}, collections.defaultdict(lambda: collections.Counter())
    for r, c in st.items():
        f = waves.get(r, alias.get(r, 0))
        n = static_mix.copies.get(r, 1)
        ex = {k: c.get(k, 0) * f / n for k in CLASSES + ["valu_cost", "issue_cost"]}
        out_regions[r] = {"phase": PHASE_OF.get(r, "other"), "wave_executions": f, "inlined_copies": n, "lanes_per_execution": round(lanes.get(r, 0) / f, 2) if f else None,
                          "static": {k: c.get(k, 0) for k in CLASSES if c.get(k, 0)}, "executed": {k: v for k, v in ex.items() if v}}
        phases[PHASE_OF.get(r, "other")].update(ex)
    rays = fq["counters"]["rays"]
    total = collections.Counter()
    for p in phases.values():
        total.update(p)
    valu_total = sum(total[k] for k in VALU_CLASSES)
    res = {
        "kernel": a.kernel, "workload": fq.get("workload"), "counters": fq["counters"],
        "method": "static instruction mix of the product build per SRT_REGION stretch (line tables) x wave-level executions of that stretch counted by a -DSRT_REGION_COUNT build; "
                  "regions tagged RARE (slow paths behind range guards, the general pow) are counted as never executed",
        "per_phase_executed_wave_instructions": {p: {k: int(v) for k, v in c.items() if v} for p, c in phases.items()},
        "per_phase_valu": {p: int(sum(c[k] for k in VALU_CLASSES)) for p, c in phases.items()},
        "per_phase_valu_cost_share": {p: round(c["valu_cost"] / max(sum(cc["valu_cost"] for cc in phases.values()), 1), 4) for p, c in phases.items()},
        "valu_cost_note": "valu_cost = executed VALU instructions weighted by their measured issue cost (full-rate fp32 / logic / shift-right / integer add = 1, half-rate classes 1.75, transcendental 3.43; isa_phase_mix.py valu_cost)",
        "per_phase_issue_cost_share": {p: round(c["issue_cost"] / max(sum(cc["issue_cost"] for cc in phases.values()), 1), 4) for p, c in phases.items()},
        "issue_cost_note": "issue_cost = every executed instruction (vector AND scalar) weighted by what one more of its kind costs the launch, in v_add_f32 units (isa_phase_mix.py issue_cost, profiles/r04_issue_cost.json)",
        "issue_cost_per_wave_ray": round(64 * sum(c["issue_cost"] for c in phases.values()) / rays, 1),
        "per_phase_valu_share": {p: round(sum(c[k] for k in VALU_CLASSES) / valu_total, 4) for p, c in phases.items()},
        "per_phase_valu_lane_slots_per_ray": {p: round(64 * sum(c[k] for k in VALU_CLASSES) / rays, 1) for p, c in phases.items()},
        "total_executed": {k: int(v) for k, v in total.items() if v},
        "total_valu": int(valu_total), "valu_lane_slots_per_ray": round(64 * valu_total / rays, 1),
        "valu_class_share": {k: round(total[k] / valu_total, 4) for k in VALU_CLASSES},
        "regions": out_regions,
    }
    if a.pmc:
        pm = json.loads(Path(a.pmc).read_text())
        k = next((v for kk, v in pm.items() if "trace" in kk), None)
        if k:
            res["pmc_check"] = {
                "SQ_INSTS_VALU": k.get("SQ_INSTS_VALU"), "model_valu": int(valu_total),
                "ratio_model_over_pmc": round(valu_total / k["SQ_INSTS_VALU"], 4) if k.get("SQ_INSTS_VALU") else None,
                "SQ_INSTS_VALU_TRANS_F32": k.get("SQ_INSTS_VALU_TRANS_F32"), "model_trans": int(total["trans"]),
                "SQ_INSTS_VALU_FMA_F64": k.get("SQ_INSTS_VALU_FMA_F64"), "model_fp64": int(total["fp64"]),
                "SQ_INSTS_VALU_CVT": k.get("SQ_INSTS_VALU_CVT"), "model_cvt": int(total["cvt"]),
                "SQ_INSTS_SALU": k.get("SQ_INSTS_SALU"), "model_salu": int(total["salu"] + total["branch"] + total["s_nop"] + total["wait"]),
                "SQ_INSTS_SMEM": k.get("SQ_INSTS_SMEM"), "model_smem": int(total["smem"]),
                "SQ_INSTS_LDS": k.get("SQ_INSTS_LDS"), "model_lds": int(total["lds"]),
            }
    txt = json.dumps(res, indent=1)
    if a.out:
        Path(a.out).write_text(txt + "\n")
    print(json.dumps({k: res[k] for k in ("per_phase_issue_cost_share", "issue_cost_per_wave_ray", "per_phase_valu_cost_share", "per_phase_valu_share", "per_phase_valu_lane_slots_per_ray", "valu_class_share", "valu_lane_slots_per_ray", "pmc_check") if k in res}, indent=1))


if __name__ == "__main__":
    main()
