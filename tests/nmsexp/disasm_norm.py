import re, sys
name = "_ZN12_GLOBAL__N_112k_nmsm_roundILi2ELi0EEEvNS_9NmsLayoutEiPhmPjPiiPKyPKjiii"
def extract(path):
    out, on = [], False
    for ln in open(path):
        if ln.startswith(name + ":"):
            on = True
            continue
        if on:
            s = ln.rstrip()
            if s.strip().startswith(".Lfunc_end"):
                break
            s = re.sub(r";.*$", "", s).rstrip()          # comments
            if not s.strip() or s.strip().startswith((".p2align", ".section", ".type", ".size")):
                continue
            out.append(s.strip())
    return out
for v in sys.argv[1:]:
    lines = extract(v + ".s")
    open(v + ".round2.s", "w").write("\n".join(lines) + "\n")
    print(v, len(lines), "lines,", sum(1 for l in lines if not l.endswith(":")), "instructions")
