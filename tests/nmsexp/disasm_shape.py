import re, sys
def shape(l):
    l = re.sub(r"v\[\d+:\d+\]", "v[]", l)
    l = re.sub(r"s\[\d+:\d+\]", "s[]", l)
    l = re.sub(r"\bv\d+\b", "v", l)
    l = re.sub(r"\bs\d+\b", "s", l)
    l = re.sub(r"\.LBB\d+_\d+", ".L", l)
    return l
for v in sys.argv[1:]:
    open(v + ".shape.s", "w").write("".join(shape(l) for l in open(v + ".round2.s")))
