#!/usr/bin/env python3
"""CPU-only check of a filed NMS round listing (tests/nmsexp/disasm/*.s) for the two things a wave can get wrong around
per-lane predicated loads (DESIGN section 4 "predicated-load hazard"):

  U  a vector register read by a lane that no instruction has written for that lane under the exec masks the wave really
     had (a value loaded under a narrow exec mask and consumed after exec was widened, without a wide initialisation);
  W  a vector register read or overwritten while a load into it is still outstanding by the s_waitcnt vmcnt bookkeeping
     (loads retire in order; stores and atomics without return are counted as the hardware counts them);
  H  a vector instruction that reads a scalar register or vcc fewer than two wait states after a vector instruction wrote it
     (the gfx940-family rule LLVM's hazard recogniser fills with s_nop; -disable-peephole moves some of those s_nop).

Method: a wave-level walk of the listing that is CONCRETE in control (exec, vcc, scalar masks and the scalar loop counters are
64-bit / 32-bit integers) and ABSTRACT in data (a vector register holds, per lane, only "defined?" and a value id).  A vector
compare yields a pseudo-random mask that is a pure function of (opcode, operand value ids, lane), so the same test of the same
values gives the same answer wherever the compiler repeats it.  Many walks with different seeds and different compare
densities cover the paths; every report carries the listing line, so it can be read against the .s file.

    python3 tests/nmsexp/exec_flow.py tests/nmsexp/disasm/round_s3_fail.s [walks]
"""
import re, sys, random, hashlib

M64 = (1 << 64) - 1

def parse(path):
    prog, labels = [], {}
    for ln, raw in enumerate(open(path), 1):
        s = raw.strip()
        if not s:
            continue
        if s.endswith(":"):
            labels[s[:-1]] = len(prog)
            continue
        parts = s.split(None, 1)
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        prog.append((ln, parts[0], ops, s))
    return prog, labels

def vregs(op):
    m = re.fullmatch(r"v(\d+)", op)
    if m:
        return [int(m.group(1))]
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", op)
    if m:
        return list(range(int(m.group(1)), int(m.group(2)) + 1))
    return []

def sreg(op):
    """-> ('s', first index, width) | ('vcc',) | ('exec',) | None"""
    m = re.fullmatch(r"s(\d+)", op)
    if m:
        return ("s", int(m.group(1)), 1)
    m = re.fullmatch(r"s\[(\d+):(\d+)\]", op)
    if m:
        return ("s", int(m.group(1)), int(m.group(2)) - int(m.group(1)) + 1)
    if op in ("vcc", "exec"):
        return (op,)
    return None

def literal(op):
    try:
        return int(op, 0)
    except ValueError:
        return {"-1": -1}.get(op)

class Wave:
    def __init__(self, prog, labels, seed, density):
        self.prog, self.labels, self.rnd, self.density, self.seed = prog, labels, random.Random(seed), density, seed
        self.s = {}                    # scalar registers: index -> 32-bit int or None
        self.vcc = 0
        self.exec = M64
        self.scc = 0
        self.vdef = {}                 # v -> 64-bit mask of lanes written
        self.vid = {}                  # v -> list of 64 value ids
        self.pending = []              # outstanding VMEM ops in issue order: (line, [dest vregs] or None)
        self.nid = 1
        self.reports = {}
        self.seen = set()
        self.swr = {}                  # scalar register index or 'vcc' -> wait states since a vector instruction wrote it
        self.vconst = {}               # v -> list of 64 known small values (only for literal moves and literal selects)
        for v in range(0, 1):          # v0 = work-item id
            self.vdef[v] = M64
            self.vid[v] = list(range(64))

    # ----- scalar values
    def get64(self, op):
        r = sreg(op)
        if r is None:
            v = literal(op)
            return None if v is None else v & M64
        if r[0] == "vcc":
            return self.vcc
        if r[0] == "exec":
            return self.exec
        for k in (r[1], r[1] + 1):          # a mask nothing wrote on this path holds garbage: pick it once
            if self.s.get(k) is None:
                self.s[k] = self.rnd.getrandbits(32)
        return self.s[r[1]] | (self.s[r[1] + 1] << 32)

    def set64(self, op, val):
        r = sreg(op)
        if r[0] == "vcc":
            self.vcc = (val or 0) & M64
        elif r[0] == "exec":
            self.exec = val & M64
        else:
            self.s[r[1]] = None if val is None else val & 0xffffffff
            self.s[r[1] + 1] = None if val is None else (val >> 32) & 0xffffffff

    def get32(self, op):
        r = sreg(op)
        if r is None:
            v = literal(op)
            return None if v is None else v & 0xffffffff
        if r[0] == "s":
            return self.s.get(r[1])
        return None

    def set32(self, op, val):
        r = sreg(op)
        if r and r[0] == "s":
            self.s[r[1]] = None if val is None else val & 0xffffffff

    # ----- vector bookkeeping
    def report(self, kind, line, text, detail):
        self.reports.setdefault((kind, line), (text, detail))

    def outstanding(self, v):
        for ln, dests in self.pending:
            if dests and v in dests:
                return ln
        return None

    def vread(self, line, text, v, mask):
        ln = self.outstanding(v)
        if ln is not None:
            self.report("W", line, text, "reads v%d while the load of line %d is outstanding" % (v, ln))
        miss = mask & ~self.vdef.get(v, 0)
        if miss:
            self.report("U", line, text, "v%d read by %d lane(s) nothing has written (exec %016x, written %016x)" % (v, bin(miss).count("1"), mask, self.vdef.get(v, 0)))

    def vwrite(self, line, text, v, mask, ids=None, from_load=False):
        if not from_load:
            ln = self.outstanding(v)
            if ln is not None:
                self.report("W", line, text, "overwrites v%d while the load of line %d is outstanding" % (v, ln))
        self.vdef[v] = self.vdef.get(v, 0) | mask
        self.vconst.pop(v, None)
        cur = self.vid.setdefault(v, [0] * 64)
        self.nid += 1
        for l in range(64):
            if mask >> l & 1:
                cur[l] = ids[l] if ids else self.nid * 64 + l

    def cmp_mask(self, mnem, ops):
        out = 0
        srcs = []
        for o in ops:
            vr = vregs(o)
            srcs.append(("v", vr) if vr else ("c", o if sreg(o) is None else repr(self.get64(o) if sreg(o)[0] != "s" or sreg(o)[2] == 2 else self.get32(o))))
        # how often this KIND of test (opcode + constant operands) is true differs from walk to walk, so that walks exist in which
        # blockers are rare while candidates are common and the other way round
        kind = repr((mnem, self.seed, [x for k, x in srcs if k == "c"])).encode()
        dens = (0.0, 0.02, 0.1, 0.3, 0.5, 0.7, 0.9, 0.98, 1.0)[(hashlib.blake2b(kind, digest_size=2).digest()[0] + int(self.density * 100)) % 9]
        for l in range(64):
            if not self.exec >> l & 1:
                continue
            key = [mnem, self.seed]
            for k, x in srcs:
                key.append(tuple(self.vid.get(v, [0] * 64)[l] for v in x) if k == "v" else x)
            h = int.from_bytes(hashlib.blake2b(repr(key).encode(), digest_size=4).digest(), "little")
            if h / 2 ** 32 < dens:
                out |= 1 << l
        return out

    # ----- one walk
    def run(self, max_steps=200000):
        pc, P = 0, self.prog
        for _ in range(max_steps):
            if pc >= len(P):
                return "fell off"
            line, m, ops, text = P[pc]
            self.seen.add(pc)
            pc += 1
            if m == "s_endpgm":
                return "end"
            ws = int(ops[0], 0) + 1 if m == "s_nop" else 1
            if not m.startswith("s_") and not m.startswith("global_") and not m.startswith("ds_"):
                self.hazard(line, m, ops, text)
            for k in list(self.swr):
                self.swr[k] += ws
                if self.swr[k] > 8:
                    del self.swr[k]
            if not m.startswith("s_"):
                self.note_swrite(m, ops)
            if m in ("s_nop", "s_sleep", "s_setprio", "s_barrier"):
                continue
            if m == "s_waitcnt":
                mm = re.search(r"vmcnt\((\d+)\)", text)
                if mm:
                    n = int(mm.group(1))
                    while len(self.pending) > n:
                        self.pending.pop(0)
                continue
            if m == "s_branch":
                pc = self.labels[ops[0]]
                continue
            if m.startswith("s_cbranch_"):
                c = {"execz": self.exec == 0, "execnz": self.exec != 0, "vccz": self.vcc == 0, "vccnz": self.vcc != 0,
                     "scc0": self.scc == 0, "scc1": self.scc == 1}[m[len("s_cbranch_"):]]
                if c:
                    pc = self.labels[ops[0]]
                continue
            if m.startswith("s_"):
                self.salu(line, m, ops, text)
                continue
            self.valu(line, m, ops, text)
        return "step limit"

    def sidx(self, op):
        r = sreg(op.split()[0]) if op.split() else None
        if r is None:
            return []
        if r[0] == "vcc":
            return ["vcc"]
        if r[0] == "s":
            return list(range(r[1], r[1] + r[2]))
        return []

    def note_swrite(self, m, ops):
        w = []
        if m.startswith("v_cmp"):
            w = self.sidx(ops[0]) if (m.endswith("_e64") or ops[0] == "vcc") else ["vcc"]
        elif m == "v_readfirstlane_b32":
            w = self.sidx(ops[0])
        elif m.startswith("v_mad_u64_u32") or m.startswith("v_mad_i64_i32"):
            w = self.sidx(ops[1])
        for k in w:
            self.swr[k] = 0

    def hazard(self, line, m, ops, text):
        srcs = ops[1:]
        if m.startswith("v_cmp") and not m.endswith("_e64") and ops and ops[0] != "vcc":
            srcs = ops
        reads = [k for o in srcs for k in self.sidx(o)]
        if m in ("v_cndmask_b32_e32",) and len(ops) < 4:
            reads.append("vcc")
        for k in reads:
            if k in self.swr and self.swr[k] < 2:
                self.report("H", line, text, "reads %s %d wait state(s) after a vector instruction wrote it" % ("vcc" if k == "vcc" else "s%d" % k, self.swr[k]))

    def salu(self, line, m, ops, text):
        b64 = {"s_and_b64": lambda a, b: a & b, "s_or_b64": lambda a, b: a | b, "s_xor_b64": lambda a, b: a ^ b,
               "s_andn2_b64": lambda a, b: a & ~b, "s_orn2_b64": lambda a, b: a | (~b & M64)}
        if m in b64:
            a, b = self.get64(ops[1]), self.get64(ops[2])
            r = None if a is None or b is None else b64[m](a, b) & M64
            if r is None and ops[0] == "exec":
                raise RuntimeError("exec from an unknown mask at line %d: %s" % (line, text))
            self.set64(ops[0], r)
            self.scc = int(bool(r)) if r is not None else self.rnd.randint(0, 1)
            return
        sx = {"s_and_saveexec_b64": lambda s, e: s & e, "s_or_saveexec_b64": lambda s, e: s | e, "s_andn2_saveexec_b64": lambda s, e: s & ~e,
              "s_xor_saveexec_b64": lambda s, e: s ^ e}
        if m in sx:
            src = self.get64(ops[1])
            if src is None:
                raise RuntimeError("saveexec from an unknown mask at line %d: %s" % (line, text))
            old = self.exec
            self.exec = sx[m](src, old) & M64
            self.set64(ops[0], old)
            self.scc = int(self.exec != 0)
            return
        if m == "s_mov_b64":
            self.set64(ops[0], self.get64(ops[1]))
            return
        if m == "s_mov_b32":
            self.set32(ops[0], self.get32(ops[1]))
            return
        if m == "s_cselect_b64":
            self.set64(ops[0], self.get64(ops[1]) if self.scc else self.get64(ops[2]))
            return
        if m in ("s_add_u32", "s_add_i32", "s_addc_u32", "s_sub_i32"):
            a, b = self.get32(ops[1]), self.get32(ops[2])
            if a is None or b is None or (m == "s_addc_u32" and self.scc is None):
                self.set32(ops[0], None)
                self.scc = self.rnd.randint(0, 1)
                return
            r = a - b if m == "s_sub_i32" else a + b + (self.scc if m == "s_addc_u32" else 0)
            self.set32(ops[0], r)
            self.scc = int(r > 0xffffffff or r < 0) if m != "s_add_i32" and m != "s_sub_i32" else 0
            return
        if m.startswith("s_cmp"):
            a, b = self.get32(ops[0]), self.get32(ops[1])
            if a is None or b is None:
                self.scc = self.rnd.randint(0, 1)
                return
            sg = lambda x: x - (1 << 32) if x >> 31 else x
            if m.startswith("s_cmpk"):
                b = b & 0xffff
                b = b - 0x10000 if (b >> 15 and "_i32" in m) else b
                b &= 0xffffffff
            op = m.split("_")[2]
            x, y = (sg(a), sg(b)) if m.endswith("i32") else (a, b)
            self.scc = int({"eq": x == y, "lg": x != y, "gt": x > y, "ge": x >= y, "lt": x < y, "le": x <= y}[op])
            return
        # everything else (s_load, s_mul, s_lshl, s_bcnt1, ...): result unknown
        if ops:
            r = sreg(ops[0])
            if r and r[0] == "s":
                for k in range(r[2]):
                    self.s[r[1] + k] = None
        if m in ("s_bcnt1_i32_b64", "s_abs_i32", "s_lshl_b32", "s_lshl_b64", "s_ashr_i32"):
            self.scc = self.rnd.randint(0, 1)

    def valu(self, line, m, ops, text):
        E = self.exec
        ops = [o.split()[0] for o in ops if o.split()]          # drop modifiers (sc0, sc1, offset:..)
        if m.startswith("global_load"):
            for v in vregs(ops[1]):
                self.vread(line, text, v, E)
            d = vregs(ops[0])
            for v in d:
                self.vwrite(line, text, v, E, from_load=True)
            self.pending.append((line, d))
            return
        if m.startswith("global_store") or m.startswith("global_atomic"):
            ret = m.startswith("global_atomic") and "sc0" in text
            for o in ops[1 if ret else 0:]:
                for v in vregs(o):
                    self.vread(line, text, v, E)
            d = vregs(ops[0]) if ret else None
            for v in d or []:
                self.vwrite(line, text, v, E, from_load=True)
            self.pending.append((line, d))
            return
        if m.startswith("v_cmp"):
            e64 = m.endswith("_e64")
            srcs = ops[1:] if e64 else ops[1:] if ops[0] == "vcc" else ops
            for o in srcs:
                for v in vregs(o):
                    self.vread(line, text, v, E)
            res = self.cmp_mask(m[:-4], srcs)
            mm = re.fullmatch(r"v_cmp_(eq|ne)_u32_e(32|64)", m)
            if mm and len(srcs) == 2 and literal(srcs[0]) is not None and vregs(srcs[1]) and vregs(srcs[1])[0] in self.vconst:
                c, vals = literal(srcs[0]), self.vconst[vregs(srcs[1])[0]]
                res = sum(1 << l for l in range(64) if E >> l & 1 and vals[l] is not None and (vals[l] == c) == (mm.group(1) == "eq"))
            self.set64(ops[0] if (e64 or ops[0] == "vcc") else "vcc", res)
            return
        if m.startswith("v_cndmask_b32"):
            sel = self.get64(ops[3]) if len(ops) > 3 else self.vcc
            if sel is None:
                sel = 0
            for o, mask in ((ops[1], E & ~sel), (ops[2], E & sel)):
                for v in vregs(o):
                    self.vread(line, text, v, mask)
            for v in vregs(ops[0]):
                old = self.vconst.get(v)
                self.vwrite(line, text, v, E)
                a, b = literal(ops[1]), literal(ops[2])
                if a is not None and b is not None:
                    self.vconst[v] = [(b if sel >> l & 1 else a) if E >> l & 1 else (old[l] if old else None) for l in range(64)]
            return
        if m == "v_readfirstlane_b32":
            for v in vregs(ops[1]):
                self.vread(line, text, v, E & -E if E else 1)
            self.set32(ops[0], None)
            return
        dests = vregs(ops[0])
        srcs = ops[1:]
        if m.startswith("v_mad_u64_u32") or m.startswith("v_mad_i64_i32"):
            self.set64(ops[1], None) if sreg(ops[1]) else None
            srcs = ops[2:]
        copy = None
        if m in ("v_mov_b32_e32", "v_mov_b64_e32") and vregs(srcs[0]):
            copy = vregs(srcs[0])
        for o in srcs:
            for v in vregs(o):
                self.vread(line, text, v, E)
        for k, v in enumerate(dests):
            self.vwrite(line, text, v, E, ids=list(self.vid.get(copy[k], [0] * 64)) if copy else None)

def walk_all(prog, labels, walks):
    """-> ({(kind, listing line): (text, detail, walk)}, instruction indices reached, {how walks ended: count})"""
    allrep, ends, seen = {}, {}, set()
    for w in range(walks):
        wave = Wave(prog, labels, seed=w, density=(0.03, 0.1, 0.3, 0.5, 0.7, 0.9, 0.97)[w % 7])
        if w % 3 == 1:
            wave.exec = random.Random(w).getrandbits(64) | 1     # partial wavefronts too
        if w % 9 == 2:
            wave.exec = 1 << (w % 64) | 1 << (w * 7 % 64)        # and nearly empty ones
        try:
            e = wave.run()
        except RuntimeError as ex:
            e = "stopped: %s" % ex
        ends[e.split(":")[0]] = ends.get(e.split(":")[0], 0) + 1
        seen |= wave.seen
        for k, v in wave.reports.items():
            allrep.setdefault(k, v + (w,))
    return allrep, seen, ends

def main():
    path = sys.argv[1]
    walks = int(sys.argv[2]) if len(sys.argv) > 2 else 420
    prog, labels = parse(path)
    allrep, seen, ends = walk_all(prog, labels, walks)
    print("%s: %d instructions, %d walks, ends %s, %d instructions reached" % (path, len(prog), walks, ends, len(seen)))
    miss = sorted(set(range(len(prog))) - seen)
    if miss:
        print("never reached (listing lines):", ", ".join(str(prog[i][0]) for i in miss[:40]), "..." if len(miss) > 40 else "")
    for kind in "UWH":
        rows = sorted((k[1], v) for k, v in allrep.items() if k[0] == kind)
        print("%s reports: %d" % (kind, len(rows)))
        for ln, (text, detail, w) in rows:
            print("  line %4d  %-60s %s  [walk %d]" % (ln, text, detail, w))

if __name__ == "__main__":
    main()
