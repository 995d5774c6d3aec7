"""CPU test: the C# side of the drop-in (integration/csharp/*.cs) is held to include/pgx.h.

There is no .NET toolchain in this image or on the GPU box, so the binding cannot be compiled; what CAN be checked is
that it says what the header says.  Every [DllImport] of PgxNative.cs is parsed (name, return type, every parameter's
type) and compared with the prototype of include/pgx.h: parameter count, and per parameter the C# type the C type
marshals to (handles as IntPtr / out IntPtr, int* as int* or `out int`, size_t as nuint, uint64_t as ulong, int64_t* as
long*, device pointers as void* or their typed form); PgxKeypoint / PgxPair against pgx_keypoint / pgx_pair field for
field; the status, stage and source-format constants against the header's values; every native that GpuStages.cs calls
must be declared, with the argument count of the call; and the exports the binding leaves out are exactly the ones it
lists as OMITTED.  The checker is itself checked: removing a parameter on either side must turn it red.

Reference call sites the binding replaces: Photogrammetry/Program.cs:42-55 (DI registrations),
ImageProcessing.Abstractions/Keypoint.cs:17-27, KeypointPair.cs:3-8.
"""
import os
import re

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
HEADER = os.path.join(ROOT, "include", "pgx.h")
NATIVE = os.path.join(ROOT, "integration", "csharp", "PgxNative.cs")
STAGES = os.path.join(ROOT, "integration", "csharp", "GpuStages.cs")


def _strip_c_comments(s):
    return re.sub(r"/\*.*?\*/", " ", s, flags=re.S)


def _strip_cs_comments(s):
    s = re.sub(r"/\*.*?\*/", " ", s, flags=re.S)
    return re.sub(r"//[^\n]*", " ", s)


def _split_args(a):
    a = a.strip()
    if not a or a == "void":
        return []
    return [x.strip() for x in a.split(",")]


def c_prototypes(text):
    """name -> (return type, [parameter types]) for every pgx_* function of the header text."""
    src = _strip_c_comments(text)
    protos = {}
    for m in re.finditer(r"([A-Za-z_][A-Za-z_0-9 \*]*?)\b(pgx_[a-z_0-9]+)\s*\(([^()]*)\)\s*;", src):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3)
        params = []
        for p in _split_args(args):
            p = re.sub(r"\s+", " ", p)
            mm = re.match(r"^(.*?)([A-Za-z_][A-Za-z_0-9]*)$", p)   # the last identifier is the parameter's name
            params.append(mm.group(1).strip() if mm and mm.group(1).strip() else p)
        protos[name] = (re.sub(r"\s+", " ", ret), params)
    return protos


def cs_imports(text):
    """name -> (return type, [parameter types]) for every [DllImport] of the C# text."""
    src = _strip_cs_comments(text)
    out = {}
    for m in re.finditer(r"\[DllImport\(Lib\)\]\s*public\s+static\s+extern\s+([A-Za-z\*]+)\s+(pgx_[a-z_0-9]+)\s*\(([^()]*)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        params = []
        for p in _split_args(args):
            toks = p.split()
            params.append(" ".join(toks[:-1]))   # drop the parameter's name
        out[name] = (ret, params)
    return out


def norm_c(t):
    t = t.replace("const", " ")
    t = re.sub(r"\s+", " ", t).strip()
    return t.replace(" *", "*").replace("* ", "*")


# what a C parameter type may be bound as
def allowed_cs(ctype):
    t = norm_c(ctype)
    scalars = {"int": {"int"}, "float": {"float"}, "double": {"double"}, "size_t": {"nuint"}, "uint64_t": {"ulong"},
               "int64_t": {"long"}, "int32_t": {"int"}, "uint32_t": {"uint"}}
    if t in scalars:
        return scalars[t]
    if t in ("pgx_ctx*", "pgx_tracks*"):
        return {"IntPtr"}
    if t in ("pgx_ctx**", "pgx_tracks**"):
        return {"out IntPtr"}
    typed = {"int*": {"int*", "out int"}, "int32_t*": {"int*", "out int"}, "uint32_t*": {"uint*"}, "uint16_t*": {"ushort*"},
             "float*": {"float*"}, "double*": {"double*"}, "int64_t*": {"long*"}, "pgx_keypoint*": {"PgxKeypoint*"},
             "pgx_pair*": {"PgxPair*"}, "void*": {"void*", "byte*", "IntPtr"}, "char*": {"byte*", "IntPtr", "string"},
             "uint32_t**": {"uint**"}}
    if t in typed:
        return typed[t] | {"void*"}     # any pointer may be bound untyped (device pointers are)
    raise AssertionError("test_csharp_binding: no rule for C type %r" % ctype)


def allowed_ret(ctype):
    t = norm_c(ctype)
    return {"int": {"int"}, "void": {"void"}, "char*": {"IntPtr"}}[t]


def compare(protos, imports):
    """List of human-readable mismatches between the header's prototypes and the C# declarations."""
    bad = []
    for name, (ret, params) in sorted(imports.items()):
        if name not in protos:
            bad.append("%s: declared in C#, not in pgx.h" % name)
            continue
        cret, cparams = protos[name]
        if ret not in allowed_ret(cret):
            bad.append("%s: returns %s, header says %s" % (name, ret, cret))
        if len(params) != len(cparams):
            bad.append("%s: %d parameters, header has %d" % (name, len(params), len(cparams)))
            continue
        for i, (cs, c) in enumerate(zip(params, cparams)):
            if cs not in allowed_cs(c):
                bad.append("%s: parameter %d is %r, header says %r" % (name, i, cs, c))
    return bad


def test_every_dllimport_matches_its_prototype():
    protos = c_prototypes(open(HEADER).read())
    imports = cs_imports(open(NATIVE).read())
    assert len(protos) >= 50 and len(imports) >= 38, (len(protos), len(imports))
    assert compare(protos, imports) == []


def test_omitted_exports_are_exactly_the_listed_ones():
    protos = c_prototypes(open(HEADER).read())
    text = open(NATIVE).read()
    imports = cs_imports(text)
    listed = set()
    for line in re.findall(r"//\s*OMITTED:([^\n]*)", text):
        listed |= set(line.split())
    assert listed, "PgxNative.cs must list the exports it leaves out (// OMITTED: ...)"
    assert set(protos) - set(imports) == listed
    assert not (listed & set(imports))


def test_struct_layouts():
    h = _strip_c_comments(open(HEADER).read())
    cs = _strip_cs_comments(open(NATIVE).read())

    def c_fields(name):
        body = re.search(r"typedef\s+struct\s*\{([^}]*)\}\s*%s\s*;" % name, h).group(1)
        out = []
        for decl in [d.strip() for d in body.split(";") if d.strip()]:
            typ, names = decl.split(None, 1)
            out += [typ] * len(names.split(","))
        return out

    def cs_fields(name):
        m = re.search(r"\[StructLayout\(LayoutKind\.Sequential\)\]\s*public\s+struct\s+%s\s*\{([^}]*)\}" % name, cs)
        assert m, "%s must be a sequential-layout struct" % name
        out = []
        for decl in [d.strip() for d in m.group(1).split(";") if d.strip()]:
            toks = decl.replace("public", "").split(None, 1)
            out += [toks[0]] * len(toks[1].split(","))
        return out
    to_cs = {"int32_t": "int", "float": "float"}
    assert [to_cs[t] for t in c_fields("pgx_keypoint")] == cs_fields("PgxKeypoint") == ["int", "int", "int", "float"]
    assert [to_cs[t] for t in c_fields("pgx_pair")] == cs_fields("PgxPair") == ["int", "int", "int"]


def test_constants_equal_the_header():
    h = _strip_c_comments(open(HEADER).read())
    cs = _strip_cs_comments(open(NATIVE).read())
    defines = {m.group(1): int(m.group(2)) for m in re.finditer(r"#define\s+(PGX_[A-Z_0-9]+)\s+(-?\d+)\b", h)}
    for m in re.finditer(r"enum\s*\{([^}]*)\}", h):
        for item in m.group(1).split(","):
            k, v = item.split("=")
            defines[k.strip()] = int(v)
    consts = {}
    for m in re.finditer(r"public\s+const\s+int\s+([^;]*);", cs):
        for item in m.group(1).split(","):
            k, v = item.split("=")
            consts[k.strip()] = v.strip()

    def key(c_name):   # PGX_E_DIM_MISMATCH -> edimmismatch
        return c_name[4:].replace("_", "").lower()
    cs_by_key = {k.lower(): v for k, v in consts.items()}
    checked = 0
    for c_name, val in defines.items():
        if c_name == "PGX_H":
            continue
        k = key(c_name)
        assert k in cs_by_key, "%s has no C# constant" % c_name
        got = cs_by_key[k]
        assert (int(got) if got.lstrip("-").isdigit() else {"int.MaxValue": 2**31 - 1}[got]) == val, c_name
        checked += 1
    assert checked >= 9 + 1 + 2 + 4 + 1   # status codes, PGX_DIST_NONE, source formats, stages, PGX_COMM_ID_BYTES


def _calls(text, prefix="PgxNative."):
    """(name, number of top-level arguments) of every prefix + pgx_*(...) call."""
    out = []
    for m in re.finditer(re.escape(prefix) + r"(pgx_[a-z_0-9]+)\s*\(", text):
        i, depth, args, cur = m.end(), 1, 0, False
        while depth:
            ch = text[i]
            if ch in "([{":
                depth += 1
            elif ch in ")]}":
                depth -= 1
            elif ch == "," and depth == 1:
                args += 1
            if depth >= 1 and not ch.isspace():
                cur = True
            i += 1
        out.append((m.group(1), args + 1 if cur else 0))
    return out


def test_every_native_the_stage_classes_call_is_declared_with_that_arity():
    imports = cs_imports(open(NATIVE).read())
    calls = _calls(_strip_cs_comments(open(STAGES).read())) + _calls(_strip_cs_comments(open(NATIVE).read()))
    assert len(calls) >= 8
    for name, nargs in calls:
        assert name in imports, "%s is called but not declared in PgxNative.cs" % name
        assert nargs == len(imports[name][1]), "%s called with %d arguments, declared with %d" % (name, nargs, len(imports[name][1]))


def test_the_checker_turns_red():
    """Removing a parameter on either side, changing a type class or a return type must be reported."""
    h, cs = open(HEADER).read(), open(NATIVE).read()
    protos, imports = c_prototypes(h), cs_imports(cs)
    assert compare(protos, imports) == []
    # C# side loses a parameter
    cs2 = cs.replace("pgx_set_detect_params(IntPtr ctx, float threshold, int suppressionRadius)", "pgx_set_detect_params(IntPtr ctx, float threshold)")
    assert cs2 != cs and any("pgx_set_detect_params" in b for b in compare(protos, cs_imports(cs2)))
    # header side loses a parameter
    h2 = h.replace("int pgx_set_detect_params(pgx_ctx *ctx, float threshold, int suppression_radius);", "int pgx_set_detect_params(pgx_ctx *ctx, float threshold);")
    assert h2 != h and any("pgx_set_detect_params" in b for b in compare(c_prototypes(h2), imports))
    # a wrong type class
    cs3 = cs.replace("pgx_allgather_dev(IntPtr ctx, void* dBuf, nuint bytesPerRank)", "pgx_allgather_dev(IntPtr ctx, void* dBuf, int bytesPerRank)")
    assert cs3 != cs and any("pgx_allgather_dev" in b for b in compare(protos, cs_imports(cs3)))
    cs4 = cs.replace("ulong seed", "int seed")
    assert cs4 != cs and any("pgx_fundamental_ransac_dev" in b for b in compare(protos, cs_imports(cs4)))
    # a wrong return type
    cs5 = cs.replace("public static extern void pgx_ctx_destroy", "public static extern int pgx_ctx_destroy")
    assert cs5 != cs and any("pgx_ctx_destroy" in b for b in compare(protos, cs_imports(cs5)))
    # a declaration the header does not have
    cs6 = cs.replace("pgx_check_status(IntPtr ctx)", "pgx_check_statusx(IntPtr ctx)")
    assert any("pgx_check_statusx" in b for b in compare(protos, cs_imports(cs6)))
