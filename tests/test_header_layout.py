"""include/jtk_lc.h compiled as plain C (what cgo / bindgen / a C host see): every struct's size and field offsets against
the ctypes / numpy mirror the tests and bench.py call the library through (jtk_amd/ffi.py).  The Python-side layout tests
compare two Python mirrors with each other; this one asks the C compiler."""
import ctypes as C
import json
import os
import subprocess

import numpy as np

from jtk_amd import ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

STRUCTS = {
    "jtk_hmm_t": (ffi.Hmm, None),
    "jtk_gain_profile_t": (ffi.GainProfile, None),
    "jtk_gains_t": (ffi.Gains, None),
    "jtk_lc_params_t": (ffi.Params, None),
    "jtk_lc_timing_t": (ffi.Timing, None),
    "jtk_lc_chunk_t": (None, ffi.CHUNK_DT),
    "jtk_lc_result_t": (None, ffi.RESULT_DT),
    "jtk_lc_feature_chunk_t": (None, ffi.FEATURE_CHUNK_DT),
    "jtk_cc_node_t": (None, ffi.CC_NODE_DT),
    "jtk_cc_chunk_t": (None, ffi.CC_CHUNK_DT),
}


def mirror_layout(ct, dt):
    if ct is not None:
        return C.sizeof(ct), {name: getattr(ct, name).offset for name, _ in ct._fields_}
    return dt.itemsize, {name: dt.fields[name][1] for name in dt.names}


def test_header_compiles_as_c_and_matches_the_python_binding(tmp_path):
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "jtk_lc.h"', 'int main(void) {', 'printf("{");']
    first = True
    for cname, (ct, dt) in STRUCTS.items():
        _, offs = mirror_layout(ct, dt)
        lines.append('printf("%s\\"%s\\": {\\"sizeof\\": %%zu", sizeof(%s));' % ("" if first else ", ", cname, cname))
        first = False
        for f in offs:
            lines.append('printf(", \\"%s\\": %%zu", offsetof(%s, %s));' % (f, cname, f))
        lines.append('printf("}");')
    lines += ['printf(", \\"K_COUNT\\": %d, \\"GAINS_MAX_HOMOP\\": %d}\\n", (int)JTK_K_COUNT, (int)JTK_GAINS_MAX_HOMOP);',
              'return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    # -std=c11 -pedantic-errors: the header must be C, not C++ that happens to compile
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-pedantic-errors", "-I", os.path.join(ROOT, "include"),
                           str(src), "-o", str(exe)])
    got = json.loads(subprocess.check_output([str(exe)]))
    assert got.pop("K_COUNT") == ffi.K_COUNT and got.pop("GAINS_MAX_HOMOP") == ffi.GAINS_MAX_HOMOP
    for cname, (ct, dt) in STRUCTS.items():
        size, offs = mirror_layout(ct, dt)
        c = got[cname]
        assert c.pop("sizeof") == size, cname
        assert c == offs, cname
    # every field of the numpy records is one the header has (no padding fields invented on the Python side)
    assert set(ffi.CHUNK_DT.names) == {"chunk_id", "copy_num", "n_reads", "tmpl_off", "tmpl_len", "read_first"}
    assert np.dtype(ffi.RESULT_DT).itemsize == 24


def test_a_plain_c_host_links_and_gets_status_codes_without_a_gpu(tmp_path, jtk_lib):
    """the boundary is a C ABI: a C11 translation unit links libjtk_lc.so, calls the entry points a host touches first and
    gets status codes (never a crash, never a CPU fallback) on a machine without a device"""
    src = tmp_path / "host.c"
    src.write_text(r'''
#include <stdio.h>
#include <string.h>
#include "jtk_lc.h"
int main(void) {
    jtk_lc_params_t p;
    memset(&p, 0, sizeof p);
    jtk_lc_result_t res;
    uint32_t label = 0;
    double post = 0.0;
    int rc_null = jtk_lc_cluster_chunks(NULL, 0, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, 1, NULL, NULL, NULL, 0,
                                        NULL, NULL, 0, 0);
    int rc_dev = jtk_lc_cluster_chunks(&p, 0, NULL, NULL, NULL, NULL, NULL, NULL, NULL, &label, &post, 1, &res, NULL, NULL, 0,
                                       NULL, NULL, 0, 99);
    char err[256];
    strncpy(err, jtk_lc_last_error(), sizeof err - 1);   /* valid until this thread's next call into the library */
    err[sizeof err - 1] = 0;
    int ok = jtk_lc_device_ok(99);
    printf("%d %d %d %s|%s\n", rc_null, rc_dev, ok, jtk_lc_strerror(rc_null), err);
    return 0;
}
''')
    exe = tmp_path / "host"
    libdir = os.path.dirname(ffi.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                           "-L", libdir, "-ljtk_lc", "-Wl,-rpath," + libdir])
    out = subprocess.check_output([str(exe)], env=dict(os.environ, LD_LIBRARY_PATH="/opt/rocm/lib")).decode()
    rc_null, rc_dev, dev_ok, rest = out.split(" ", 3)
    assert int(rc_null) < 0 and int(rc_dev) < 0 and int(dev_ok) == 0, out     # errors, not results: no device ordinal 99
    assert rest.split("|")[0].strip() != "" and rest.split("|")[1].strip() != ""
