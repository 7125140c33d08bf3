"""Layout gate for the Rust binding (bindings/rust, delivered UN-COMPILED: no rustc in this image).

`src/ffi.rs` is parsed and held against include/rttnw_hip.h — every struct (field order, names, types), every entry
point (name, arity, argument and return types) — and against the ctypes binding the GPU tests exercise (sizes, field
offsets).  Adding a field to `rttnw_params`, or an export to the header, without touching ffi.rs fails here.
"""
import ctypes as C
import os
import re

from rttnw_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = open(os.path.join(ROOT, "include", "rttnw_hip.h")).read()
FFI = open(os.path.join(ROOT, "bindings", "rust", "src", "ffi.rs")).read()


def strip_c(text):
    return re.sub(r"/\*.*?\*/", "", text, flags=re.S)


def strip_rs(text):
    return re.sub(r"//[^\n]*", "", text)


C_TO_RS = {"uint32_t": "u32", "uint64_t": "u64", "int32_t": "i32", "uint8_t": "u8", "double": "f64", "int": "c_int", "void": "c_void",
           "char": "c_char", "rttnw_id": "rttnw_id"}
RS_CTYPES = {"u32": C.c_uint32, "u64": C.c_uint64, "i32": C.c_int32, "f64": C.c_double, "u8": C.c_uint8}


def c_type_to_rs(t):
    """'const double*' -> '*const f64', 'rttnw_scene**' -> '*mut *mut rttnw_scene', 'uint32_t' -> 'u32'."""
    t = t.strip()
    const = t.startswith("const ")
    if const:
        t = t[6:].strip()
    stars = t.count("*")
    base = t.replace("*", "").strip()
    base = C_TO_RS.get(base, base)
    if stars == 0:
        return base
    out = base
    for k in range(stars):
        out = ("*const " if (const and k == 0) else "*mut ") + out
    return out


def header_structs():
    out = {}
    for body, name in re.findall(r"typedef struct \w+ \{(.*?)\}\s*(\w+);", strip_c(HEADER), flags=re.S):
        fields = []
        for decl in body.split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            if "(*" in decl:                                   # function pointer member of the builder table
                fields.append((re.search(r"\(\*(\w+)\)", decl).group(1), "fnptr"))
                continue
            m = re.match(r"(.+?)\s+([\w, \[\]]+)$", decl)
            ctype, names = m.group(1), m.group(2)
            for nm in names.split(","):
                nm = nm.strip()
                arr = re.match(r"(\w+)\[(\d+)\]", nm)
                if arr:
                    fields.append((arr.group(1), "[%s; %s]" % (c_type_to_rs(ctype), arr.group(2))))
                else:
                    fields.append((nm, c_type_to_rs(ctype)))
        out[name] = fields
    return out


def rust_structs():
    out = {}
    for attrs, name, body in re.findall(r"((?:#\[[^\]]*\]\s*)+)pub struct (\w+)\s*\{(.*?)\n\}", strip_rs(FFI), flags=re.S):
        assert "repr(C)" in attrs, name
        fields = []
        for nm, ty in re.findall(r"(?:pub\s+)?(\w+)\s*:\s*([^,\n]+(?:\([^)]*\)[^,\n]*)?),", body):
            fields.append((nm, "fnptr" if "extern \"C\" fn" in ty else ty.strip()))
        out[name] = fields
    return out


def header_functions():
    text = strip_c(HEADER)
    text = re.sub(r"typedef struct \w+ \{.*?\}\s*\w+;", "", text, flags=re.S)   # drop struct bodies (function-pointer members)
    text = re.sub(r"^\s*#.*$", "", text, flags=re.M)                              # preprocessor lines
    text = re.sub(r"enum \w+ \{.*?\};", "", text, flags=re.S)
    text = re.sub(r"typedef [^;{]*;", "", text)
    funcs = {}
    for ret, name, args in re.findall(r"([\w\s\*]+?)\s*\b(rttnw_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        ret = " ".join(ret.split())
        if ret.startswith("typedef") or "(" in ret:
            continue
        arg_types = []
        args = " ".join(args.split())
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                a = re.sub(r"\[\d*\]$", "*", a)                                   # `const double center[3]` -> pointer
                m = re.match(r"(.+?[\s\*])(\w+)\*?$", a)
                ctype = m.group(1).strip() + ("*" if a.endswith("*") else "")
                arg_types.append(c_type_to_rs(ctype))
        funcs[name] = (arg_types, c_type_to_rs(ret))
    return funcs


def rust_functions():
    block = re.search(r'extern "C" \{(.*)\n\}', strip_rs(FFI), flags=re.S).group(1)
    funcs = {}
    for name, args, ret in re.findall(r"pub fn (\w+)\((.*?)\)\s*(?:->\s*([^;]+))?;", block, flags=re.S):
        arg_types = [a.split(":", 1)[1].strip() for a in args.split(",") if ":" in a]
        funcs[name] = (arg_types, (ret or "c_void").strip())
    return funcs


def test_every_struct_matches_the_header_field_for_field():
    hs, rs = header_structs(), rust_structs()
    assert set(hs) == {"rttnw_camera_desc", "rttnw_params", "rttnw_stats", "rttnw_tile_layout", "rttnw_build_info", "rttnw_builder_api"}
    for name, fields in hs.items():
        assert name in rs, "ffi.rs lacks struct %s" % name
        assert rs[name] == fields, (name, rs[name], fields)


def test_struct_layouts_match_the_ctypes_binding_the_gpu_tests_use():
    pairs = {"rttnw_camera_desc": abi.CameraDesc, "rttnw_params": abi.Params, "rttnw_stats": abi.Stats,
             "rttnw_tile_layout": abi.TileLayout, "rttnw_build_info": abi.BuildInfo}
    rs = rust_structs()
    for name, ct in pairs.items():
        offset = 0
        assert [f[0] for f in ct._fields_] == [f[0] for f in rs[name]], name
        for (fname, rty), (cname, cty) in zip(rs[name], ct._fields_):
            arr = re.match(r"\[(\w+); (\d+)\]", rty)
            want = RS_CTYPES[arr.group(1)] * int(arr.group(2)) if arr else RS_CTYPES[rty]
            assert C.sizeof(want) == C.sizeof(cty), (name, fname)
            align = C.alignment(want)
            offset = (offset + align - 1) // align * align                      # #[repr(C)] layout rule
            assert offset == getattr(ct, cname).offset, (name, fname, offset)
            offset += C.sizeof(want)
        align = max(C.alignment(t) for _, t in ct._fields_)
        assert (offset + align - 1) // align * align == C.sizeof(ct), name
    assert C.sizeof(abi.Params) == 88 and C.sizeof(abi.CameraDesc) == 120 and C.sizeof(abi.Stats) == 64


def test_every_export_of_the_header_is_bound_with_the_same_signature():
    hf, rf = header_functions(), rust_functions()
    assert len(hf) >= 38
    assert set(hf) == set(rf), (sorted(set(hf) - set(rf)), sorted(set(rf) - set(hf)))
    for name, (args, ret) in hf.items():
        r_args, r_ret = rf[name]
        assert r_args == args and r_ret == ret, (name, r_args, args, r_ret, ret)
    # and the ctypes binding used by the tests names the same exports
    assert set(abi.exported_symbols()) <= set(hf)


def test_builder_table_order_matches_the_ctypes_table():
    assert [f[0] for f in rust_structs()["rttnw_builder_api"]] == [n for n, _, _ in abi.BUILDER_FUNCS]


def test_constants_match():
    for name, val in re.findall(r"#define (RTTNW_\w+) (\d+)u?\b", strip_c(HEADER)):
        m = re.search(r"pub const %s: \w+ = (\w+);" % name, FFI)
        assert m, name
        assert m.group(1) == val or m.group(1).startswith("RTTNW_"), name
    for enum_body in re.findall(r"enum rttnw_\w+ \{(.*?)\}", strip_c(HEADER), flags=re.S):
        nxt = 0
        for item in enum_body.split(","):
            item = item.strip()
            if not item:
                continue
            if "=" in item:
                nm, v = [x.strip() for x in item.split("=")]
                nxt = int(v)
            else:
                nm = item
            m = re.search(r"pub const %s: \w+ = (-?\d+);" % nm, FFI)
            assert m and int(m.group(1)) == nxt, nm
            nxt += 1
