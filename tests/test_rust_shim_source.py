"""The Rust binding a jtk maintainer adds (rust/*.rs; not compilable here: no cargo/rustc in the image) is kept in step
with include/jtk_lc.h: every extern fn exists in the header with the same number of parameters, and every #[repr(C)]
struct lists the header's fields in the header's order."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def strip_comments(text, line="//"):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return "\n".join(l.split(line)[0] for l in text.splitlines())


def c_functions(header):
    out = {}
    for m in re.finditer(r"\b(jtk_lc_[a-z_0-9]+)\s*\(([^;{]*?)\)\s*;", header, flags=re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ("", "void") else len(args.split(","))
    return out


def c_struct_fields(header, name):
    body = re.search(r"typedef struct %s \{(.*?)\} %s_t;" % (name, name), header, flags=re.S).group(1)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        names = decl.split(None, 1)[1] if not decl.startswith("jtk_") else decl.split(None, 1)[1]
        for n in names.split(","):
            fields.append(re.sub(r"\[.*\]", "", n).strip().lstrip("*"))
    return fields


def rust_struct_fields(src, name):
    body = re.search(r"pub struct %s \{(.*?)\}" % name, src, flags=re.S).group(1)
    return re.findall(r"pub ([a-z_0-9]+)\s*:", body)


def test_rust_ffi_matches_the_header():
    header = strip_comments(open(os.path.join(ROOT, "include", "jtk_lc.h")).read())
    rust = strip_comments(open(os.path.join(ROOT, "rust", "gpu_ffi.rs")).read())
    cf = c_functions(header)
    ext = re.search(r'extern "C" \{(.*)\}', rust, flags=re.S).group(1)
    seen = 0
    for m in re.finditer(r"pub fn (jtk_lc_[a-z_0-9]+)\s*\((.*?)\)\s*(?:->\s*[^;]+)?;", ext, flags=re.S):
        name, args = m.group(1), m.group(2).strip()
        assert name in cf, name
        assert (len([a for a in args.split(",") if a.strip()]) if args else 0) == cf[name], name
        seen += 1
    assert seen >= 7 and "jtk_lc_cluster_chunks" in ext
    for c_name, r_name in [("jtk_hmm", "JtkHmm"), ("jtk_gain_profile", "JtkGainProfile"), ("jtk_gains", "JtkGains"),
                           ("jtk_lc_params", "JtkLcParams"), ("jtk_lc_chunk", "JtkLcChunk"), ("jtk_lc_result", "JtkLcResult"),
                           ("jtk_cc_node", "JtkCcNode"), ("jtk_cc_chunk", "JtkCcChunk")]:
        assert rust_struct_fields(rust, r_name) == c_struct_fields(header, c_name), c_name


def test_shim_sources_are_present_and_cite_the_reference():
    for f, needle in [("build.rs", "rustc-link-lib=dylib=jtk_lc"), ("gpu_shim.rs", "jtk_lc_cluster_chunks"),
                      ("gpu_shim.rs", "mod.rs:244-260")]:
        assert needle in open(os.path.join(ROOT, "rust", f)).read(), (f, needle)
