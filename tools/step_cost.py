#!/usr/bin/env python3
"""Instruction classes of the tile kernels' step loops and their issue cost with the measured class rates
(profiles/r03a_valu_calibration.json): compiles sls_render_block.hip to assembly (no GPU needed) and counts the inner
loops of the production instantiations (forward <8,2,false,true>, backward <8,2,true,true,0>).
    python tools/step_cost.py [extra hipcc flags]"""
import collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "splat_loam_amd", "csrc", "sls_render_block.hip")
FLAGS = ("--offload-arch=gfx950 -O3 -std=c++17 -fPIC -DSLS_TILE_W=16 -DSLS_TILE_H=16 "
         "-fhip-fp32-correctly-rounded-divide-sqrt -munsafe-fp-atomics -fno-slp-vectorize -S --cuda-device-only").split()
CLK = {"plain": 2.3, "cndmask": 4.2, "v_cmp": 4.2, "dpp": 4.2, "packed": 4.2, "transc": 8.2, "permlane": 8.2, "readlane": 4.2}


def classify(op):
    if op.startswith("v_cndmask"): return "cndmask"
    if op.startswith("v_cmp"): return "v_cmp"
    if "dpp" in op: return "dpp"
    if op.startswith(("v_exp", "v_rcp", "v_sqrt", "v_rsq", "v_log")): return "transc"
    if op.startswith("v_permlane"): return "permlane"
    if op.startswith("v_pk_"): return "packed"
    if op.startswith(("v_readlane", "v_readfirstlane")): return "readlane"
    if op.startswith("v_"): return "plain"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_")): return "vmem"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith(("s_cbranch", "s_branch")): return "branch"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_"): return "salu"
    return "other"


def main():
    out = "/tmp/step_cost.s"
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + sys.argv[1:] + [SRC, "-o", out], stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
    for name, prefix, marker in (("forward step", "_ZN3sls23render_fwd_dense_kernelILi8ELi2ELb0ELb1EE", "v_exp_f32"),
                                 ("backward step", "_ZN3sls23render_bwd_block_kernelILi8ELi2ELb1ELi2ELi0ELb1EE", "global_atomic_add_f32")):
        a = [i for i, l in enumerate(lines) if l.startswith(prefix)][0]
        b = [i for i in range(a, len(lines)) if "s_endpgm" in lines[i]][0]
        seg = lines[a:b]
        hs = [i for i, l in enumerate(seg) if "Inner Loop Header: Depth=2" in l]
        ms = [i for i, l in enumerate(seg) if marker in l]
        # the step loop: the depth-2 loop that holds the marker instruction (a peeled first iteration may precede it)
        m = [i for i in ms if any(h < i for h in hs)][0 if name.startswith("backward") else -1]
        h = max(x for x in hs if x < m)
        hdr = [l for l in seg[h - 3:h + 1] if l.startswith(".LBB")][-1].split(":")[0].lstrip(".L")
        # the loop = the header's block + every block annotated "in Loop: Header=<hdr> Depth=2" (they may sit before it)
        cnt = collections.Counter()
        inside = False
        for i, l in enumerate(seg):
            if l.startswith(".LBB"):
                inside = (f"Header={hdr} Depth=2" in l) or (l.split(":")[0].lstrip(".L") == hdr)
                continue
            if "Inner Loop Header: Depth=2" in l and i == h:
                inside = True
            t = l.strip()
            if not inside or not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
                continue
            cnt[classify(t.split()[0])] += 1
        valu = {k: v for k, v in cnt.items() if k in CLK}
        clocks = sum(CLK[k] * v for k, v in valu.items())
        print(f"{name}: VALU {sum(valu.values())} {dict(sorted(valu.items()))}  issue clocks {clocks:.0f};  "
              f"salu {cnt['salu']} branch {cnt['branch']} lds {cnt['lds']} waitcnt {cnt['waitcnt']} nop {cnt['nop']} vmem {cnt['vmem']}")
    vg = [l for l in lines if "NumVgprs" in l or ("; ScratchSize" in l)]
    print("VGPRs / scratch of the instantiations:", " ".join(x.split(":")[1].strip() for x in vg[:24]))


if __name__ == "__main__":
    main()
