"""The C-ABI shared library: loads, exports every symbol include/*.h declares,
agrees with the ctypes mirror on struct layout, and fails LOUDLY (status 4, no CPU fallback)
when asked to compute without a HIP device.  No compute calls are made when a GPU is absent.
"""
from __future__ import annotations

import ctypes as C
import re
import subprocess
import tempfile
from pathlib import Path

import numpy as np
import pytest

import scenes

ROOT = Path(__file__).resolve().parent.parent
HEADER = ROOT / "include" / "f3d_terrain_pt.h"


@pytest.fixture(scope="module")
def native():
    import __graft_entry__ as g

    g.build_hip()
    from forge3d_amd import _native

    return _native


def _declared_functions():
    text = "\n".join(h.read_text() for h in sorted((ROOT / "include").glob("*.h")))
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(f3d_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(native):
    L = native.lib()
    declared = _declared_functions()
    assert len(declared) >= 16
    for name in declared:
        assert hasattr(L, name), f"libf3dhip.so does not export {name}"
    assert sorted(n for n, _, _ in native.ABI) == declared  # the ctypes table covers the whole header
    out = subprocess.run(["nm", "-D", "--defined-only", str(native.library_path())], capture_output=True, text=True)
    exported = {line.split()[-1] for line in out.stdout.splitlines() if " T " in line}
    assert set(declared) <= exported


def test_struct_layouts_match_the_header(native):
    src = r'''
#include <stdio.h>
#include <stddef.h>
#include "f3d_terrain_pt.h"
int main(void) {
  printf("%zu %zu %zu\n", sizeof(f3d_terrain_ref_desc), sizeof(f3d_terrain_ref_out), sizeof(f3d_session_opts));
  printf("%zu %zu %zu %zu\n", offsetof(f3d_terrain_ref_desc, observer_latitude_deg),
         offsetof(f3d_terrain_ref_desc, env_map), offsetof(f3d_terrain_ref_desc, variance_threshold),
         offsetof(f3d_terrain_ref_out, gpu_resource_bytes));
  return 0; }
'''
    with tempfile.TemporaryDirectory() as tmp:
        c = Path(tmp) / "layout.c"
        c.write_text(src)
        exe = Path(tmp) / "layout"
        subprocess.run(["gcc", "-I", str(ROOT / "include"), str(c), "-o", str(exe)], check=True)
        sizes, offsets = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines()
    assert [int(x) for x in sizes.split()] == [C.sizeof(native.Desc), C.sizeof(native.Out), C.sizeof(native.SessionOpts)]
    assert [int(x) for x in offsets.split()] == [native.Desc.observer_latitude_deg.offset, native.Desc.env_map.offset,
                                                 native.Desc.variance_threshold.offset,
                                                 native.Out.gpu_resource_bytes.offset]


def _rust_structs_of_integration_md():
    """{Rust struct name: [(field, rust type)]} of every #[repr(C)] struct in INTEGRATION.md."""
    import re

    text = (ROOT / "INTEGRATION.md").read_text()
    out = {}
    for m in re.finditer(r"#\[repr\(C\)\]\s*pub struct (\w+) \{(.*?)\n\}", text, re.S):
        body = re.sub(r"//[^\n]*", "", m.group(2))
        out[m.group(1)] = [(f, t.strip()) for f, t in re.findall(r"pub (\w+):\s*([^,]+?),", body + ",")]
    return out


_RUST_TO_C = {"u32": "uint32_t", "i32": "int32_t", "f32": "float", "f64": "double", "u64": "uint64_t", "u16": "uint16_t", "u8": "uint8_t"}


def _c_decl(field, rust):
    import re

    m = re.fullmatch(r"\[(\w+);\s*(\d+)\]", rust)
    if m:
        return f"{_RUST_TO_C[m.group(1)]} {field}[{m.group(2)}];"
    m = re.fullmatch(r"\*(const|mut) (\w+)", rust)
    if m:
        return f"const void *{field};"  # every data pointer has one size and alignment
    return f"{_RUST_TO_C[rust]} {field};"


def test_integration_md_rust_structs_match_the_header():
    """The binding INTEGRATION.md shows must be the binding that works (round-2 verdict: the Rust struct lacked
    `atmosphere`): its #[repr(C)] field lists, transcribed to C, have the header's size and the header's offset for
    EVERY field."""
    rust = _rust_structs_of_integration_md()
    pairs = {"F3dTerrainRefDesc": "f3d_terrain_ref_desc", "F3dTerrainRefOut": "f3d_terrain_ref_out", "F3dAetherLuts": "f3d_aether_luts",
             "F3dAetherRefDesc": "f3d_aether_ref_desc", "F3dAetherRefOut": "f3d_aether_ref_out"}
    assert set(pairs) <= set(rust), sorted(rust)
    assert rust["F3dTerrainRefDesc"][0][0] == "struct_size" and rust["F3dTerrainRefDesc"][-1][0] == "atmosphere"
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "f3d_terrain_pt.h"']
    for rname in pairs:
        lines.append(f"typedef struct {{ {' '.join(_c_decl(f, t) for f, t in rust[rname])} }} md_{rname};")
    lines.append("int main(void) { int bad = 0;")
    for rname, cname in pairs.items():
        lines.append(f'  if (sizeof(md_{rname}) != sizeof({cname})) {{ printf("size of {rname}: %zu vs %zu\\n", sizeof(md_{rname}), sizeof({cname})); bad++; }}')
        for f, _ in rust[rname]:
            lines.append(f'  if (offsetof(md_{rname}, {f}) != offsetof({cname}, {f})) {{ printf("{rname}.{f}: %zu vs %zu\\n", '
                         f'offsetof(md_{rname}, {f}), offsetof({cname}, {f})); bad++; }}')
    lines.append("  return bad; }")
    with tempfile.TemporaryDirectory() as tmp:
        c = Path(tmp) / "md_layout.c"
        c.write_text("\n".join(lines))
        exe = Path(tmp) / "md_layout"
        subprocess.run(["gcc", "-I", str(ROOT / "include"), str(c), "-o", str(exe)], check=True)
        run = subprocess.run([str(exe)], capture_output=True, text=True)
    assert run.returncode == 0, run.stdout
    text = (ROOT / "INTEGRATION.md").read_text()
    assert "f3d_abi_version" in text and "F3D_ABI_VERSION: u32 = 6" in text


def test_abi_version_and_struct_size_are_enforced(native):
    """A caller built against another revision of the header (round 1 -> 2 grew the descriptor by a pointer) is refused
    with a value error before the library reads a member."""
    L = native.lib()
    assert L.f3d_abi_version() == native.ABI_VERSION == 6
    header = (ROOT / "include" / "f3d_terrain_pt.h").read_text()
    assert "#define F3D_ABI_VERSION 6u" in header
    dem = np.zeros((4, 4), np.float32)
    desc, keep = native.make_desc(dem, 8, 8, {}, (1.0, 1.0), 1.0, (0.6, 0.6, 0.6), 315.0, 45.0, 2.5, None, 0.35, None, None, 1, 2, 2,
                                  1e30, 7, (1.0, 1.0, 1.0), 0.0, 0.0, "ellipsoid", 6371008.8, "bennett", 0.13, 1013.25, 15.0)
    assert desc.struct_size == C.sizeof(native.Desc)
    out = native.Out()
    err = C.create_string_buffer(512)
    for wrong in (0, C.sizeof(native.Desc) - 8, C.sizeof(native.Desc) + 8):  # unset / the round-1 struct / a future one
        desc.struct_size = wrong
        assert L.f3d_terrain_ref_render(C.byref(desc), C.byref(out), err, 512) == native.STATUS_VALUE
        assert b"struct_size" in err.value and b"another revision" in err.value
    desc.struct_size = C.sizeof(native.Desc)
    opts = native.SessionOpts()  # struct_size left 0
    handle = C.c_void_p(None)
    assert L.f3d_session_create(C.byref(desc), C.byref(opts), C.byref(handle), err, 512) == native.STATUS_VALUE
    assert b"f3d_session_opts.struct_size" in err.value and not handle.value


def test_a_wavefront_scene_of_the_previous_header_revision_is_still_read(native):
    """f3d_wf_scene has only grown at its end (round 5: + primary_start): a caller whose struct ends in front of the new member
    gets past the size check (and then fails for want of a device here, status 4), any other size is refused (status 1)."""
    from forge3d_amd import wavefront as w

    s, keep = w._marshal(w.adjudication_scene().wavefront_scene().as_dict())
    hdr, rgba, acc = np.zeros((8, 8, 4), np.float32), np.zeros((8, 8, 4), np.uint8), np.zeros((8, 8, 4), np.float32)
    out = w._Out(hdr.ctypes.data, rgba.ctypes.data, acc.ctypes.data, 0.0, 0, 0)
    err = C.create_string_buffer(512)

    def call(size):
        s.struct_size = size
        return native.lib().f3d_wavefront_render(C.byref(s), 8, 8, 0, 1, 1, 0, C.byref(out), err, 512)

    full, previous = C.sizeof(w._Scene), w._Scene.primary_start.offset
    assert previous == full - 8
    for size in (full, previous):
        rc = call(size)
        assert rc != native.STATUS_VALUE or b"struct_size" not in err.value
    for size in (0, previous - 8, full + 8):
        assert call(size) == native.STATUS_VALUE and b"struct_size" in err.value


def test_version_and_device_probe(native):
    L = native.lib()
    assert b"gfx950" in L.f3d_version()
    assert L.f3d_device_count() >= 0


def test_host_side_earth_model_through_the_abi(native):
    """f3d_effective_radius_m is pure host arithmetic (src/geo/refraction.rs): no GPU needed."""
    from oracle import oracle

    L = native.lib()
    out = C.c_double(0.0)
    err = C.create_string_buffer(256)
    for earth, refr, az in ((2, 1, 225.0), (2, 0, 0.0), (1, 2, 90.0), (2, 3, 302.0)):
        rc = L.f3d_effective_radius_m(earth, 12.5, 6371008.8, refr, 1013.25, 15.0, 0.13, az, C.byref(out), err, 256)
        assert rc == 0
        want = oracle.effective_radius_m({1: "sphere", 2: "ellipsoid"}[earth],
                                         {0: "none", 1: "bennett", 2: "saemundsson", 3: "effective_radius"}[refr], az,
                                         latitude_deg=12.5)
        assert out.value == want
    assert L.f3d_effective_radius_m(0, 0.0, 6371008.8, 1, 1013.25, 15.0, 0.13, 0.0, C.byref(out), err, 256) == 2
    assert b"flat earth" in err.value


def test_compute_entry_points_fail_loudly_without_a_gpu(native):
    if native.device_count() > 0:
        pytest.skip("a HIP device is present; the no-device behaviour is checked in the CPU container")
    import forge3d_amd

    dem = scenes.golden_dem(4)
    with pytest.raises(RuntimeError, match=r"\[Device\].*no CPU fallback"):
        forge3d_amd.hybrid_render_terrain_reference(dem, 32, 32, scenes.CAM, max_frames=4, min_frames=2)
    L = native.lib()
    err = C.create_string_buffer(256)
    hit = np.zeros(1, np.uint32)
    rays = np.zeros((1, 8), np.float32)
    rc = L.f3d_terrain_trace_batch(dem.ctypes.data, dem.shape[1], dem.shape[0], 0.0, 0.0, 1.0, 1.0, 1.0, 0.0, 0,
                                   rays.ctypes.data, 1, 1, 1, hit.ctypes.data, None, None, err, 256)
    assert rc == native.STATUS_DEVICE and b"no CPU fallback" in err.value
    tot = C.c_uint64(0)
    rc = L.f3d_build_minmax_mips(dem.ctypes.data, dem.shape[1], dem.shape[0], None, None, 0, C.byref(tot), err, 256)
    assert rc == -native.STATUS_DEVICE


def test_product_package_never_imports_the_oracle():
    """The oracle and the emulator are test infrastructure: nothing under forge3d_amd/ may
    reference them."""
    forbidden = re.compile(r"(import\s+oracle|from\s+oracle|from\s+emul|import\s+emul|#include\s+\"[^\"]*(oracle|emul)"
                           r"|libf3d_oracle|libf3d_emul|f3do_|emul_render|emul_session)")
    checked = 0
    for path in (ROOT / "forge3d_amd").rglob("*"):
        if path.suffix in (".py", ".h", ".hip", ".cpp"):
            checked += 1
            hit = forbidden.search(path.read_text())
            assert hit is None, f"{path} references test infrastructure: {hit.group(0)!r}"
    assert checked >= 10
    # and the shared library links nothing but the HIP runtime / libstdc++
    out = subprocess.run(["ldd", str(ROOT / "forge3d_amd" / "libf3dhip.so")], capture_output=True, text=True).stdout
    assert "oracle" not in out and "emul" not in out


def test_the_library_says_which_sources_it_was_built_from():
    """f3d_source_digest: __graft_entry__.build_hip stamps the SHA-256 of csrc/ + include/ + the compiler flags into
    the library, _native.lib() refuses a library whose stamp differs from the tree, build_hip rebuilds exactly then."""
    import __graft_entry__ as entry
    from forge3d_amd import _native

    digest = _native.source_digest()
    assert digest and len(digest) == 16
    assert _native.lib().f3d_source_digest().decode() == digest
    assert entry._built_digest(entry.LIB) == digest  # so build() has nothing to do


def test_python_and_library_agree_on_the_halo():
    from forge3d_amd import _native
    from forge3d_amd.distributed import HALO_ROWS, RES_BYTES
    from forge3d_amd.session import HALO_ROWS as session_rows, RESERVOIR_BYTES, reservoir_buffer_bytes

    assert _native.lib().f3d_halo_rows() == HALO_ROWS == session_rows == 4
    assert RES_BYTES == RESERVOIR_BYTES == 16
    assert reservoir_buffer_bytes(10, 7) == (10 + 2 * 4) * 7 * 16


def test_kernel_variant_fields_have_names():
    """f3d_session_opts.kernel_variant packs its A/B switches as decimal fields; the package builds and reads them by name."""
    from forge3d_amd.session import describe_kernel_variant, kernel_variant

    assert kernel_variant() == 0
    assert kernel_variant(sample_lanes=4, waves_per_simd=5) == 4000105
    assert kernel_variant(sample_lanes=8, tile_map=1) == 8001000
    for v in (0, 4000105, 8001000, 160105, 640000000 + 4000000):
        assert kernel_variant(**describe_kernel_variant(v)) == v
    with pytest.raises(ValueError):
        kernel_variant(sample_lanes=3)
