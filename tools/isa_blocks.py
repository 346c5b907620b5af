#!/usr/bin/env python3
"""Per-basic-block instruction census of one kernel in a hipcc -S listing (no GPU needed).
usage: python tools/isa_blocks.py LISTING.s KERNEL_SUBSTRING [MIN_INSTRUCTIONS] [--dump LABEL]
Prints, for every basic block with at least MIN_INSTRUCTIONS (default 30): MFMA, other VALU, v_accvgpr copies, LDS, waits, nops."""
import re, sys
from collections import Counter

def main():
    path, key = sys.argv[1], sys.argv[2]
    args = [a for a in sys.argv[3:] if not a.startswith("--")]
    mn = int(args[0]) if args else 30
    dump = sys.argv[sys.argv.index("--dump") + 1] if "--dump" in sys.argv else None
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*%s\S*:" % re.escape(key), l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    print(lines[start].split(":")[0])
    blocks, cur = [], ["entry"]
    blocks.append(cur)
    for l in lines[start + 1:end]:
        if re.match(r"^\.LBB\d+_\d+:", l):
            cur = [l.split(":")[0]]
            blocks.append(cur)
        else:
            cur.append(l)
    for bb in blocks:
        ins = [x.strip().split()[0] for x in bb[1:] if x.strip() and not x.strip().startswith((";", "."))]
        if dump and bb[0] == dump:
            print("\n".join(bb))
        if len(ins) < mn:
            continue
        c = Counter(ins)
        pick = lambda p: sum(v for k, v in c.items() if k.startswith(p))
        print(f"{bb[0]:>10}: n={len(ins)} mfma={pick('v_mfma')} valu={pick('v_') - pick('v_mfma') - pick('v_accvgpr')} accvgpr={pick('v_accvgpr')} "
              f"ds={pick('ds_')} vmem={pick('global_') + pick('buffer_') + pick('scratch_')} salu={pick('s_') - c.get('s_waitcnt', 0) - c.get('s_nop', 0)} "
              f"waitcnt={c.get('s_waitcnt', 0)} nop={c.get('s_nop', 0)}")
        print("            ", dict(c.most_common(12)))

if __name__ == "__main__":
    main()
