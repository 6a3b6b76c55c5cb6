"""Host-side surface: settings.config grammar, environment generator / image loader, text writers.
CPU only.  The parser is compared field by field with the REFERENCE's own parser where
oracle/_ref has been built (this container); the generator with the oracle restatement."""
import os
import subprocess

import numpy as np
import pytest

import host_lib
import maps

CONFIGS = {
    "shipped_like": """# Mode\nmode=1\nncols=1000\nnrows=1000\nnb_of_obstacles=15\nminWidth=100\nmaxWidth=200\nminHeight=100\nmaxHeight=200\nrandomSeed=1\nseedValue=25\nimagePath=path\\images\\maze_6.png\nstart={50,50}\nend={990,990}\nmax_iter=250\nvisibilityThreshold=0.25\nlightStrength=1\ntimer=1\nsaveResults=1\nsaveCameFrom=1\nsaveLightSources=1\nsaveGlobalVisibility=1\nsaveLocalVisibility=1\nsaveVisibilityField=1\nsilent=0\nballRadius=15\n""",
    "spaces_tabs_bools": "mode = 2 \n\tncols\t=\t77\nnrows= 33\nrandomSeed=false\nsilent=true\ntimer=false\nstart = {3,4}\nend={ 5,6}\nvisibilityThreshold=1\n",
    "unknown_and_comments": "#c\n\nfoo=bar\n # not a comment = but has equals\nmode=7\nsilent=1\nnoequals\nempty=\nballRadius=-3\nseedValue=-12\n",
    "bad_pair": "silent=1\nstart=(1,2)\nend={7}\n",
    "lenient_minheight": "silent=1\nminHeight=abc\nmax_iter=xyz\nnrows=12\n",
    "strict_ncols": "silent=1\nncols=abc\nnrows=12\n",
    "negative_size": "silent=1\nmaxWidth=-4\n",
    "bad_bool": "silent=1\nsaveCameFrom=yes\n",
    "threshold_range": "silent=1\nvisibilityThreshold=1.5\n",
    "light_range": "silent=1\nlightStrength=-0.1\n",
    "nb_negative_wraps": "silent=1\nnb_of_obstacles=-1\n",
    "crlf": "silent=1\r\nncols=12\r\n",
    "value_with_equals": "silent=1\nimagePath=a=b=c.png\n",
}


@pytest.fixture(scope="module")
def host():
    host_lib.build()
    return host_lib.load()


@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_parser_matches_reference(host, tmp_path, name, capfd):
    path = tmp_path / "settings.config"
    path.write_bytes(CONFIGS[name].encode())
    ok, cfg = host_lib.parse_config(host, str(path))
    mine = capfd.readouterr()
    ref = host_lib.parse_config_reference(str(path))
    if ref is None:
        pytest.skip("oracle/_ref not built (reference tree absent)")
    theirs = capfd.readouterr()
    assert (ok, cfg) == ref
    assert mine.err == theirs.err and mine.out == theirs.out  # same diagnostics, same banner


def test_parser_shipped_file(host, capfd):
    path = "/root/reference/config/settings.config"
    if not os.path.exists(path):
        pytest.skip("reference tree absent")
    got = host_lib.parse_config(host, path)
    mine = capfd.readouterr()
    ref = host_lib.parse_config_reference(path)
    theirs = capfd.readouterr()
    assert got == ref and got[0] and got[1]["ncols"] == 1000 and got[1]["max_iter"] == 250
    assert mine.out == theirs.out


def test_parser_known_values(host, tmp_path):
    # independent of the reference build: defaults and a few values (parser.h:11-37)
    path = tmp_path / "c.config"
    path.write_text("silent=1\nstart={3,4}\nmode=2\n")
    ok, cfg = host_lib.parse_config(host, str(path))
    assert ok and cfg["mode"] == 2 and (cfg["start_x"], cfg["start_y"]) == (3, 4)
    assert cfg["ncols"] == 100 and cfg["max_iter"] == 100 and cfg["visibilityThreshold"] == 0.5 and cfg["ballRadius"] == 5
    assert not host_lib.parse_config(host, str(tmp_path / "missing.config"))[0]


@pytest.mark.parametrize("args", [(1000, 1000, 15, 100, 200, 100, 200, 1), (101, 77, 25, 2, 20, 3, 9, 42),
                                  (64, 300, 40, 1, 1, 1, 400, 7)])
def test_environment_generator(host, oracle, args):
    occ = host_lib.generate_env(host, *args)
    assert occ.tobytes() == oracle.generate_env(*args).tobytes()


def test_environment_density_kat(host):
    import platform
    if platform.libc_ver()[1] != "2.35":
        pytest.skip("glibc rand() stream differs")
    occ = host_lib.generate_env(host, 1000, 1000, 15, 100, 200, 100, 200, 1)
    assert "%.4f" % (100.0 * (occ == 0).sum() / occ.size) == "20.3243"


def test_image_loader(host, tmp_path):
    want = maps.maze_6()
    ref_png = "/root/reference/images/maze_6.png"
    if os.path.exists(ref_png):
        assert host_lib.load_image(host, ref_png).tobytes() == want.tobytes()
    # round trip through our own encoder: free <=> red channel == 255
    ny, nx = want.shape
    rgba = np.zeros((ny, nx, 4), np.uint8)
    rgba[..., 0] = np.where(want == 1, 255, 254)
    rgba[..., 1] = 7
    rgba[..., 3] = 255
    p = str(tmp_path / "m.png")
    assert host.vhp_host_save_png(p.encode(), rgba.ctypes.data, nx, ny) == 0
    assert host_lib.load_image(host, p).tobytes() == want.tobytes()
    from PIL import Image
    assert np.array_equal(np.array(Image.open(p)), rgba)  # a third-party decoder agrees with the encoder
    assert host_lib.load_image(host, str(tmp_path / "nope.png")) is None


def test_text_writer_formatting(host, tmp_path):
    vals = np.array([[0.0, 1.0, 0.5, 1 / 3, 2 / 3, 1e-5, 1.23456789e-7, 0.999999949, 0.9999996, 123456.7, 1e15, 5e-324]])
    vals = np.concatenate([vals, vals[:, ::-1] * 0.1])
    p = str(tmp_path / "out" / "m.txt")
    assert host.vhp_host_write_matrix_f64(p.encode(), vals.ctypes.data, vals.shape[1], vals.shape[0], 0) == 0
    assert open(p).read() == host_lib.format_matrix(vals)
    assert host.vhp_host_write_matrix_f64(p.encode(), vals.ctypes.data, vals.shape[1], vals.shape[0], 1) == 0
    assert open(p).read() == host_lib.format_matrix(vals, flip=True)
    lab = np.array([[0, 1, 63, 1000000000000000], [2, 2, 2, 2]], np.uint64)
    assert host.vhp_host_write_matrix_u64(p.encode(), lab.ctypes.data, 4, 2, 0) == 0
    assert open(p).read() == "0 1 63 1000000000000000 \n2 2 2 2 \n"


def test_abi_exports():
    """libvhp_hip.so loads without a GPU and exports every symbol include/vhp.h declares."""
    import re
    import vhp_amd
    vhp_amd.build_library()
    lib = vhp_amd.load_library()
    header = open(os.path.join(host_lib.ROOT, "include", "vhp.h")).read()
    declared = set(re.findall(r"\b(vhp_[a-z_0-9]+)\s*\(", header))
    assert declared == set(vhp_amd.ABI_SYMBOLS)
    for sym in declared:
        getattr(lib, sym)
    # no HIP device here: creation must fail loudly, never fall back to a CPU path
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(vhp_amd.VhpError):
            vhp_amd.Context(0)
    nm = subprocess.run(["nm", "-D", vhp_amd.LIB_PATH], capture_output=True, text=True).stdout
    assert "vhp_oracle" not in nm  # the product never links the oracle


def test_reconstruct_path_rejects_inconsistent_tables(oracle):
    """vhp_reconstruct_path is host code: it runs here.  A planner result of the oracle walks to the same path and
    length; labels outside the pivot list, unlabelled cells and label cycles are errors, never out-of-bounds reads."""
    import vhp_amd
    import maps
    vhp_amd.build_library()
    occ = maps.random_rect_map(60, 50, 10, 3, 12, 3, 12, 2)
    pts = maps.free_sources(occ, 2, 9)
    start, end = tuple(int(v) for v in pts[0]), tuple(int(v) for v in pts[1])
    r = oracle.solve(occ, start, end, 0.3, 40)
    assert r["status"] == 0
    d, path = vhp_amd.reconstruct_path(r["came_from"], r["pivots"], end)
    dw, pathw = oracle.reconstruct_path(r["came_from"], r["pivots"], end)
    assert d == dw and path.tolist() == pathw.tolist()
    bad = r["came_from"].copy()
    bad[end[1], end[0]] = r["n_pivots"] + 5          # label beyond the pivot list
    with pytest.raises(vhp_amd.VhpError):
        vhp_amd.reconstruct_path(bad, r["pivots"], end)
    bad[end[1], end[0]] = vhp_amd.UNLABELLED         # unlabelled end cell
    with pytest.raises(vhp_amd.VhpError):
        vhp_amd.reconstruct_path(bad, r["pivots"], end)
    # a 2-cycle between two pivots' cells: the walk is bounded by the pivot count
    came = np.full((8, 8), vhp_amd.UNLABELLED, np.uint64)
    piv = np.array([[1, 1], [5, 5], [7, 7]], np.int32)
    came[1, 1], came[5, 5], came[7, 7] = 1, 0, 1
    with pytest.raises(vhp_amd.VhpError):
        vhp_amd.reconstruct_path(came, piv, (7, 7))
