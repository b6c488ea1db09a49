"""The Rust binding cannot be compiled in this image (no rustc), so it is checked MECHANICALLY against the C header instead
(the reference's own boundary is its #[repr(C)] public types: /root/reference/src/lib.rs:15-17,79-81,122-124,
165-167,181).  Every `fn bn254_*` declared in an `extern "C"` block of bindings/rust/src/lib.rs and in the Rust snippets of
INTEGRATION.md is parsed and compared with its declaration in include/bn254_hip.h: name, arity, argument ORDER, pointer depth and
const-ness per level, and the integer kinds (usize <-> size_t, c_long <-> long, c_int <-> int, i32 <-> int32_t, u8 <-> uint8_t,
u64 <-> uint64_t, f64 <-> double); G1/G2/Gt/Fr <-> bn_g1/bn_g2/bn_gt/bn_fr; the opaque handles (bn254_ctx, bn254_multi, bn254_g2_prepared) <-> c_void.
The enum GpuOption is compared with BN254_OPT_*, the exchange kinds used by MultiGpu::with_exchange with BN254_EXCHANGE_*.
The checker itself is tested: a swapped argument pair, a dropped `const`, a wrong integer kind and a wrong discriminant must all be
reported."""
import pathlib
import re

import pytest

ROOT = pathlib.Path(__file__).resolve().parents[1]
HEADER = ROOT / "include" / "bn254_hip.h"
RUST_LIB = ROOT / "bindings" / "rust" / "src" / "lib.rs"
INTEGRATION = ROOT / "INTEGRATION.md"

C_BASE = {"void": "void", "bn254_ctx": "void", "bn254_multi": "void", "bn254_g2_prepared": "void", "bn254_multi_prepared": "void", "bn_g1": "g1", "bn_g2": "g2", "bn_gt": "gt", "bn_fr": "fr",
          "bn_ell_coeffs": "ell", "uint8_t": "u8", "int32_t": "i32", "uint64_t": "u64", "size_t": "usize", "int": "int", "long": "long",
          "double": "f64", "char": "char"}
RUST_BASE = {"c_void": "void", "G1": "g1", "G2": "g2", "Gt": "gt", "Fr": "fr", "EllCoeffs": "ell", "u8": "u8", "i32": "i32", "u64": "u64",
             "usize": "usize", "c_int": "int", "c_long": "long", "f64": "f64", "c_double": "f64", "c_char": "char"}


def _strip_c_comments(text):
    return re.sub(r"/\*.*?\*/", " ", text, flags=re.S)


def parse_c_type(t):
    """'const bn_g1 *' -> ('g1', ('const',)); 'bn254_multi **' -> ('void', ('mut', 'mut')); 'size_t' -> ('usize', ())
    pointer levels are listed from the OUTERMOST pointer inwards; each says whether what it points to is const"""
    t = t.strip()
    depth = t.count("*")
    core = t.replace("*", " ").split()
    const_base = "const" in core
    names = [w for w in core if w not in ("const", "struct", "unsigned")]
    assert len(names) == 1, f"cannot parse C type {t!r}"
    base = C_BASE[names[0]]
    if depth == 0:
        return base, ()
    # `const T **` (never used in the header) would need per-level parsing; a `const` before the base qualifies the INNERMOST pointee
    levels = ["mut"] * depth
    if const_base:
        levels[-1] = "const"
    return base, tuple(levels)


def parse_rust_type(t):
    """'*const G1' -> ('g1', ('const',)); '*mut *mut c_void' -> ('void', ('mut', 'mut'))"""
    t = t.strip()
    levels = []
    while t.startswith("*"):
        m = re.match(r"\*(const|mut)\s+", t)
        assert m, f"cannot parse Rust type {t!r}"
        levels.append(m.group(1))
        t = t[m.end():]
    t = t.split("::")[-1]
    return RUST_BASE[t], tuple(levels)


def c_declarations(text=None):
    text = _strip_c_comments(text if text is not None else HEADER.read_text())
    out = {}
    for m in re.finditer(r"^\s*([A-Za-z_][\w \t]*?[\s\*]+)(bn254_\w+)\s*\(([^)]*)\)\s*;", text, re.M):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        params = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                mm = re.match(r"^(.*?)(\w+)$", a)           # the last word is the parameter name
                params.append((mm.group(2), parse_c_type(mm.group(1))))
        out[name] = {"ret": parse_c_type(ret), "params": params}
    return out


def rust_declarations(text):
    out = {}
    for m in re.finditer(r"\bfn\s+(bn254_\w+)\s*\(([^)]*)\)\s*(?:->\s*([^;{]+?))?\s*;", text):
        name, args, ret = m.group(1), m.group(2).strip(), m.group(3)
        params = []
        for a in [x for x in args.split(",") if x.strip()]:
            pname, ptype = a.split(":", 1)
            params.append((pname.strip(), parse_rust_type(ptype)))
        out[name] = {"ret": parse_rust_type(ret) if ret else ("void", ()), "params": params}
    return out


def rust_blocks_of_markdown(text):
    return "\n".join(re.findall(r"```rust\n(.*?)```", text, re.S))


def compare(c_decls, rust_decls, where):
    """list of human-readable mismatches (empty = the binding matches the header)"""
    bad = []
    for name, r in rust_decls.items():
        c = c_decls.get(name)
        if c is None:
            bad.append(f"{where}: {name} is not declared in bn254_hip.h"); continue
        if c["ret"] != r["ret"]:
            bad.append(f"{where}: {name} returns {r['ret']} but the header says {c['ret']}")
        if len(c["params"]) != len(r["params"]):
            bad.append(f"{where}: {name} takes {len(r['params'])} arguments, the header {len(c['params'])}"); continue
        for i, ((cn, ct), (rn, rt)) in enumerate(zip(c["params"], r["params"])):
            if ct != rt:
                bad.append(f"{where}: {name} argument {i} ({rn}: {rt}) differs from the header's ({cn}: {ct})")
    return bad


def c_enum(prefix, text=None):
    text = _strip_c_comments(text if text is not None else HEADER.read_text())
    return {m.group(1): int(m.group(2)) for m in re.finditer(r"\b" + prefix + r"(\w+)\s*=\s*(-?\d+)", text)}


def camel_to_upper_snake(s):
    return re.sub(r"(?<!^)(?=[A-Z])", "_", s).upper()


def rust_option_enum(text):
    m = re.search(r"pub enum GpuOption\s*\{(.*?)\}", text, re.S)
    assert m, "GpuOption not found"
    return {camel_to_upper_snake(k): int(v) for k, v in re.findall(r"(\w+)\s*=\s*(-?\d+)", m.group(1))}


def option_mismatches(c_opts, rust_opts):
    c_opts = {k: v for k, v in c_opts.items() if not k.endswith("_")}          # BN254_OPT_COUNT_ is not an option
    bad = [f"GpuOption::{k} = {v}, header BN254_OPT_{k} = {c_opts.get(k)}" for k, v in rust_opts.items() if c_opts.get(k) != v]
    bad += [f"BN254_OPT_{k} has no GpuOption variant" for k in c_opts if k not in rust_opts]
    return bad


# ------------------------------------------------------------------------------------------------------------------- the checks
def test_header_parses_completely():
    decls = c_declarations()
    from bn_amd import _native
    assert set(decls) == set(_native.SIGNATURES), set(decls) ^ set(_native.SIGNATURES)
    assert decls["bn254_pairing_batch"]["params"] == [("ctx", ("void", ("mut",))), ("p", ("g1", ("const",))), ("q", ("g2", ("const",))),
                                                      ("out", ("gt", ("mut",))), ("n", ("usize", ()))]
    assert decls["bn254_multi_create"]["params"][2] == ("out", ("void", ("mut", "mut")))
    assert decls["bn254_ctx_destroy"]["ret"] == ("void", ()) and decls["bn254_error_string"]["ret"] == ("char", ("const",))


def test_rust_crate_matches_header():
    rust = rust_declarations(RUST_LIB.read_text())
    assert len(rust) >= 30, sorted(rust)
    assert compare(c_declarations(), rust, "bindings/rust/src/lib.rs") == []
    # every function the crate's wrappers call is declared in its extern block
    txt = RUST_LIB.read_text()
    called = set(re.findall(r"\b(bn254_\w+)\s*\(", txt))
    assert called <= set(rust), called - set(rust)
    # what the wrappers use of the crate exists there under that name and kind (checked against the reference when it is mounted: the
    # compiler is not available, a grep is): G1 / G2 `zero()` and `one()` are methods of the trait `Group`, which must then be imported;
    # Fr::zero / Gt::one are inherent
    imported = set(re.search(r"use bn::\{([^}]*)\}", txt).group(1).replace(" ", "").split(","))
    assert {"Fr", "G1", "G2", "Gt"} <= imported
    if re.search(r"\bG[12]::(zero|one)\(", txt):
        assert "Group" in imported, "G1::zero() / G2::one() need `use bn::Group`"
    ref = pathlib.Path("/root/reference/src/lib.rs")
    if ref.exists():
        rtxt = ref.read_text()
        assert re.search(r"pub trait Group", rtxt) and "impl Group for G1" in rtxt and "impl Group for G2" in rtxt
        for inherent in ("pub fn zero() -> Self { Fr(", "pub fn one() -> Self { Gt("):
            assert inherent in rtxt, inherent


def test_integration_md_snippets_match_header():
    rust = rust_declarations(rust_blocks_of_markdown(INTEGRATION.read_text()))
    assert len(rust) >= 14, sorted(rust)
    assert compare(c_declarations(), rust, "INTEGRATION.md") == []


def test_option_and_exchange_discriminants():
    txt = RUST_LIB.read_text()
    assert option_mismatches(c_enum("BN254_OPT_"), rust_option_enum(txt)) == []
    ex = c_enum("BN254_EXCHANGE_")
    m = re.search(r"None => (-?\d+), Some\(false\) => (-?\d+), Some\(true\) => (-?\d+)", txt)
    assert m and [int(x) for x in m.groups()] == [ex["AUTO"], ex["PEER"], ex["RCCL"]]
    # the ctypes mirror carries the same numbers
    from bn_amd import _native
    assert {k.upper(): v for k, v in _native.OPTIONS.items()} == {k: v for k, v in c_enum("BN254_OPT_").items() if not k.endswith("_")}
    assert {k.upper(): v for k, v in _native.EXCHANGE.items()} == ex
    # record sizes and the coefficient count the wrappers carry as constants
    sizes = {m.group(1): int(m.group(2)) for m in re.finditer(r"#define BN254_(\w+?)(?:_WIRE_BYTES)? (\d+)", HEADER.read_text())}
    consts = {m.group(1): int(m.group(2)) for m in re.finditer(r"pub const (\w+): usize = (\d+);", txt)}
    assert consts == {"PREPARED_COEFFS": sizes["PREPARED_COEFFS"], "PREPARED_NATIVE_LINES": sizes["PREPARED_NATIVE_LINES"], "PREPARED_NATIVE_BYTES": sizes["PREPARED_NATIVE_BYTES"],
                      "FR_WIRE_BYTES": sizes["FR"], "G1_WIRE_BYTES": sizes["G1"], "G2_WIRE_BYTES": sizes["G2"]}, (consts, sizes)
    # ... and the device code agrees with the header on the native table's shape (exported without a GPU)
    import ctypes as C
    lib = _native.lib()
    lib.bn254_native_table_bytes_B.restype = C.c_size_t; lib.bn254_native_table_bytes_B.argtypes = [C.c_size_t]
    assert lib.bn254_native_lines_B() == sizes["PREPARED_NATIVE_LINES"] and lib.bn254_native_table_bytes_B(3) == (3 + 1) * sizes["PREPARED_NATIVE_BYTES"]
    # the binding's own EllCoeffs mirrors bn_ell_coeffs: three arrays of 8 u64, in the header's order
    m = re.search(r"pub struct EllCoeffs \{([^}]*)\}", txt)
    assert m and re.findall(r"pub (\w+): \[u64; 8\]", m.group(1)) == re.search(r"typedef struct \{ uint64_t ([^;]*); \} bn_ell_coeffs", HEADER.read_text()).group(1).replace("[8]", "").split(", ")


def test_ctypes_table_matches_header():
    """bn_amd/_native.py SIGNATURES (what every Python caller and test goes through): arity and argument kinds"""
    import ctypes as C
    from bn_amd import _native
    decls = c_declarations()
    for name, argtypes in _native.SIGNATURES.items():
        params = decls[name]["params"]
        assert len(argtypes) == len(params), name
        for at, (pn, (base, levels)) in zip(argtypes, params):
            if levels:
                ok = at in (C.c_void_p, C.c_char_p) or hasattr(at, "_type_")          # any pointer
            else:
                ok = {"usize": C.c_size_t, "int": C.c_int, "long": C.c_long, "u64": C.c_uint64}[base] is at
            assert ok, f"{name}: argument {pn} is {base}{levels} in the header but {at} in _native.SIGNATURES"


@pytest.mark.parametrize("mutate, expect", [
    (lambda s: s.replace("fn bn254_pairing_product(ctx: *mut c_void, p: *const G1, q: *const G2, n: usize, out: *mut Gt)",
                         "fn bn254_pairing_product(ctx: *mut c_void, p: *const G1, q: *const G2, out: *mut Gt, n: usize)"), "bn254_pairing_product argument 3"),
    (lambda s: s.replace("fn bn254_g1_mul_batch(ctx: *mut c_void, p: *const G1,", "fn bn254_g1_mul_batch(ctx: *mut c_void, p: *mut G1,"), "bn254_g1_mul_batch argument 1"),
    (lambda s: s.replace("fn bn254_ctx_set_option(ctx: *mut c_void, key: c_int, value: c_long)", "fn bn254_ctx_set_option(ctx: *mut c_void, key: c_int, value: c_int)"), "bn254_ctx_set_option argument 2"),
    (lambda s: s.replace("fn bn254_g2_decode_batch(ctx: *mut c_void, bytes: *const u8, out: *mut G2, status: *mut i32, n: usize)",
                         "fn bn254_g2_decode_batch(ctx: *mut c_void, bytes: *const u8, out: *mut G1, status: *mut i32, n: usize)"), "bn254_g2_decode_batch argument 2"),
    (lambda s: s.replace("fn bn254_multi_destroy(m: *mut c_void);", "fn bn254_multi_destroy(m: *mut c_void) -> c_int;"), "bn254_multi_destroy returns"),
    (lambda s: s.replace("fn bn254_gt_inverse_batch(ctx: *mut c_void, a: *const Gt, out: *mut Gt, n: usize)", "fn bn254_gt_inverse_batch(ctx: *mut c_void, a: *const Gt, out: *mut Gt)"), "bn254_gt_inverse_batch takes 3"),
])
def test_checker_catches_deliberate_errors(mutate, expect):
    txt = RUST_LIB.read_text()
    bad_txt = mutate(txt)
    assert bad_txt != txt, "the mutation did not apply: the binding's text changed, update this test"
    found = compare(c_declarations(), rust_declarations(bad_txt), "mutated")
    assert any(expect in f for f in found), found


def test_checker_catches_a_wrong_discriminant():
    txt = RUST_LIB.read_text().replace("GtPowMode = 5", "GtPowMode = 6")
    assert any("GT_POW_MODE" in b for b in option_mismatches(c_enum("BN254_OPT_"), rust_option_enum(txt)))
    hdr = HEADER.read_text().replace("BN254_OPT_PIPELINE_SLOTS = 11", "BN254_OPT_PIPELINE_SLOTS = 14")
    assert any("PIPELINE_SLOTS" in b for b in option_mismatches(c_enum("BN254_OPT_", hdr), rust_option_enum(RUST_LIB.read_text())))
