"""The C++ host mirror of the reference's scene API (cuda_ray_tracer_amd/host/*.h) and the CLI:
compiled with g++ against libcrt_hip.so and run on CPU (everything up to the device launch)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "cuda_ray_tracer_amd")


def _build(src, out):
    cmd = ["g++", "-O1", "-std=c++17", "-Wall", "-Wextra", src, "-o", out, f"-L{PKG}", "-lcrt_hip",
           f"-Wl,-rpath,{PKG}", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.run(cmd, check=True)


def test_host_classes_flatten_and_bvh(tmp_path, crt):
    exe = str(tmp_path / "host_api_check")
    _build(os.path.join(ROOT, "tests", "host_api_check.cpp"), exe)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.strip().endswith("OK")


def test_header_is_plain_c_and_host_entry_points_work(tmp_path, crt):
    """include/crt_hip.h through a C99 compiler (what a cgo / JNI / N-API binding would see), linked against the library."""
    exe = str(tmp_path / "c_abi_check")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", os.path.join(ROOT, "tests", "c_abi_check.c"), "-o", exe,
                    f"-L{PKG}", "-lcrt_hip", f"-Wl,-rpath,{PKG}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip() == "OK", (r.returncode, r.stdout, r.stderr)


def test_cli_builds_and_rejects_unknown_arguments(crt):
    subprocess.run(["make", "-C", os.path.join(PKG, "csrc"), "-s", "../rt_main"], check=True)
    exe = os.path.join(PKG, "rt_main")
    r = subprocess.run([exe, "--bogus"], capture_output=True, text=True)
    assert r.returncode == 1 and "Unknown argument: --bogus" in r.stderr  # main_experimental.cu:86-89


@pytest.mark.gpu
def test_cli_ppm_matches_oracle(crt, orc):
    """rt_main --width 64 --samples 10 renders 32 spp (rounded up like main_experimental.cu:72);
    the PPM on stdout must equal the oracle's writer applied to the oracle's image."""
    exe = os.path.join(PKG, "rt_main")
    assert os.path.exists(exe), "build it on the CPU box first (make -C cuda_ray_tracer_amd/csrc ../rt_main)"
    for scene, world in (("small", "bvh"), ("small", "list"), ("random", "bvh")):
        r = subprocess.run([exe, "--width", "64", "--samples", "10", "--seed", "5", "--scene", scene, "--world", world],
                           capture_output=True, timeout=120)
        assert r.returncode == 0, r.stderr.decode()
        err = r.stderr.decode()
        assert "Rendering width 64 image with 32x8 blocks." in err and "Buffer creation took" in err
        osc = orc.scene_small() if scene == "small" else orc.scene_random(18, 11)
        nodes = orc.bvh_build(osc, 17)[0] if world == "bvh" else None
        cam = orc.camera(64, 32)
        sums, _ = orc.render(osc, cam, 5, 0, 32, 50, nodes, nthreads=8)
        want = orc.write_ppm(orc.resolve(sums, 32))
        assert r.stdout == want
        if scene == "random":
            # the device list of the C++ boundary: three contexts sharing this box's one GPU, tile shares
            env = dict(os.environ, CRT_DEVICES="0,0,0", CRT_SPLIT="tiles")
            r3 = subprocess.run([exe, "--width", "64", "--samples", "10", "--seed", "5", "--scene", scene, "--world", world],
                                capture_output=True, timeout=120, env=env)
            assert r3.returncode == 0, r3.stderr.decode()
            assert r3.stdout == want
            env = dict(os.environ, CRT_DEVICES="0,0")
            r2 = subprocess.run([exe, "--width", "64", "--samples", "10", "--seed", "5", "--scene", scene, "--world", "list"],
                                capture_output=True, timeout=120, env=env)
            assert r2.returncode == 0, r2.stderr.decode()
    bad = subprocess.run([exe, "--width", "32", "--samples", "1", "--gpus", "64"], capture_output=True, timeout=60)
    assert bad.returncode == 1 and b"Invalid --gpus" in bad.stderr
    none = subprocess.run([exe, "--width", "32", "--samples", "1", "--gpu", "99"], capture_output=True, timeout=60)
    assert none.returncode == 1 and b"FAILED - ABORTING" in none.stderr  # cudaCheckErrors behaviour, main_experimental.cu:13-23
