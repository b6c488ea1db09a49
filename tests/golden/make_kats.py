#!/usr/bin/env python3
"""Extract the known-answer VECTORS (decimal inputs / expected outputs only) that the
reference's own unit tests hold for the pairing hot path, and store them as JSON data.

Run in the build container only (needs /root/reference); the output
tests/golden/reference_kats.json is committed and is what travels to the GPU box.

Sources (reference file:line):
  test_str              src/fields/mod.rs:67-71
  fq12_test_vector      src/fields/mod.rs:83-169
  test_cyclotomic_exp   src/fields/mod.rs:171-201
  test_miller_loop      src/groups/mod.rs:522-547
  test_prepared_g2      src/groups/mod.rs:637-762
  test_reduced_pairing  src/groups/mod.rs:773-796
Only the numbers are kept (no source text).
"""
import json, re, sys, pathlib

REF = pathlib.Path("/root/reference/src")

def fn_body(text, name):
    m = re.search(r"^fn %s\(\) \{$" % re.escape(name), text, re.M)
    assert m, name
    end = text.index("\n}\n", m.end())
    return text[m.end():end]

def nums(body):
    return re.findall(r'from_str\("(\d+)"\)', body)

def main():
    fields = (REF / "fields/mod.rs").read_text()
    groups = (REF / "groups/mod.rs").read_text()
    out = {}

    n = nums(fn_body(fields, "test_str"))
    assert len(n) == 2
    out["test_str"] = {"minus_one_fr": n[0], "minus_one_fq": n[1]}

    n = nums(fn_body(fields, "fq12_test_vector"))
    assert len(n) == 24
    out["fq12_test_vector"] = {"start": n[:12], "finally": n[12:]}

    n = nums(fn_body(fields, "test_cyclotomic_exp"))
    assert len(n) == 24
    out["test_cyclotomic_exp"] = {"orig": n[:12], "expected": n[12:]}

    n = nums(fn_body(groups, "test_miller_loop"))
    assert len(n) == 14
    out["test_miller_loop"] = {"k1": n[0], "k2": n[1], "expected": n[2:]}

    n = nums(fn_body(groups, "test_reduced_pairing"))
    assert len(n) == 14
    out["test_reduced_pairing"] = {"k1": n[0], "k2": n[1], "expected": n[2:]}

    n = nums(fn_body(groups, "test_prepared_g2"))
    assert len(n) == 1 + 4 + 102 * 6, len(n)
    co = n[5:]
    # each coefficient is written in the order ell_0(c0,c1), ell_vw(c0,c1), ell_vv(c0,c1)
    body = fn_body(groups, "test_prepared_g2")
    first = body[body.index("EllCoeffs {"):]
    first = first[:first.index("}")]
    order = re.findall(r"(ell_0|ell_vw|ell_vv):", first)
    assert order == ["ell_0", "ell_vw", "ell_vv"], order
    out["test_prepared_g2"] = {
        "k2": n[0], "q_x": n[1:3], "q_y": n[3:5],
        "coeffs": [{"ell_0": co[6*i:6*i+2], "ell_vw": co[6*i+2:6*i+4], "ell_vv": co[6*i+4:6*i+6]}
                   for i in range(102)],
    }
    dst = pathlib.Path(__file__).with_name("reference_kats.json")
    dst.write_text(json.dumps(out, indent=1) + "\n")
    print("wrote", dst, dst.stat().st_size, "bytes")
    write_consts()


def limb_arrays(body):
    """every `[a, b, c, d]` u64-limb literal (hex or decimal) in a piece of reference text -> list of 4-int lists"""
    res = []
    for m in re.finditer(r"\[\s*((?:0x[0-9a-fA-F]+|\d+)\s*,\s*(?:0x[0-9a-fA-F]+|\d+)\s*,\s*(?:0x[0-9a-fA-F]+|\d+)\s*,\s*(?:0x[0-9a-fA-F]+|\d+))\s*\]", body):
        res.append([int(x.strip(), 0) for x in m.group(1).split(",")])
    return res


def any_fn_body(text, name):
    m = re.search(r"fn %s\([^)]*\)[^{]*\{" % re.escape(name), text)
    assert m, name
    depth, i = 1, m.end()
    while depth:
        depth += {"{": 1, "}": -1}.get(text[i], 0)
        i += 1
    return text[m.end():i]


def write_consts():
    """Numeric constants of the path as the reference spells them (Montgomery u64 limbs, little-endian).
    Used only to CHECK the independently derived constants (oracle/gen_consts.py, tools/gen_device_constants.py)."""
    fp = (REF / "fields/fp.rs").read_text()
    fq2 = (REF / "fields/fq2.rs").read_text()
    fq6 = (REF / "fields/fq6.rs").read_text()
    fq12 = (REF / "fields/fq12.rs").read_text()
    groups = (REF / "groups/mod.rs").read_text()
    c = {}
    for name in ("Fr", "Fq"):
        m = re.search(r"field_impl!\(\s*%s,(.*?)\);" % name, fp, re.S)
        arrs = limb_arrays(m.group(1))
        inv = re.findall(r"(0x[0-9a-f]+)\s*$", m.group(1).strip())
        assert len(arrs) == 4 and len(inv) == 1
        c[name] = {"modulus": arrs[0], "rsquared": arrs[1], "rcubed": arrs[2], "one": arrs[3], "inv": int(inv[0], 16)}
    c["fq_non_residue"] = limb_arrays(any_fn_body(fq2, "fq_non_residue"))[0]
    c["fq2_nonresidue"] = limb_arrays(any_fn_body(fq2, "fq2_nonresidue"))
    def frob(text, name):
        body = any_fn_body(text, name)
        arms = re.split(r"\n\s*(\d) =>", body)
        out = {}
        for k in range(1, len(arms) - 1, 2):
            out[arms[k]] = limb_arrays(arms[k + 1])
        return out
    c["fq6_frobenius_coeffs_c1"] = frob(fq6, "frobenius_coeffs_c1")
    c["fq6_frobenius_coeffs_c2"] = frob(fq6, "frobenius_coeffs_c2")
    c["fq12_frobenius_coeffs_c1"] = frob(fq12, "frobenius_coeffs_c1")
    g1p = groups[groups.index("impl GroupParams for G1Params"):groups.index("pub type G1 ")]
    g2p = groups[groups.index("impl GroupParams for G2Params"):groups.index("pub type G2 ")]
    c["g1_one_y"] = limb_arrays(any_fn_body(g1p, "one"))[0]
    c["g1_coeff_b"] = limb_arrays(any_fn_body(g1p, "coeff_b"))[0]
    c["g2_one_xy"] = limb_arrays(any_fn_body(g2p, "one"))
    c["g2_coeff_b"] = limb_arrays(any_fn_body(g2p, "coeff_b"))
    c["two_inv"] = limb_arrays(any_fn_body(groups, "two_inv"))[0]
    c["ate_loop_count"] = limb_arrays(any_fn_body(groups, "ate_loop_count"))[0]
    c["twist_mul_by_q_x"] = limb_arrays(any_fn_body(groups, "twist_mul_by_q_x"))
    c["twist_mul_by_q_y"] = limb_arrays(any_fn_body(groups, "twist_mul_by_q_y"))
    c["exp_by_neg_z_u"] = limb_arrays(any_fn_body(fq12, "exp_by_neg_z"))[0]
    dst = pathlib.Path(__file__).with_name("reference_consts.json")
    dst.write_text(json.dumps(c, indent=1) + "\n")
    print("wrote", dst, dst.stat().st_size, "bytes")

if __name__ == "__main__":
    sys.exit(main())
