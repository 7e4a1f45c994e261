"""ThreadSanitizer over the tree search's threaded host walk (round-3 advice: the race after Heap::update).  The host side of
oak_amd/csrc/search_host.hip is rebuilt with -fsanitize=thread (device code untouched; every other symbol comes from the product
library) and oakgpu_heap_selftest -- grow a random tree with the search's own threaded resolve phase, promote a child, grow on --
runs under it on 1 / 2 / 4 / 8 / 16 threads.  CPU only: no kernel is launched."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang"

DRIVER = r"""
#include <stdint.h>
#include <stdio.h>
int oakgpu_heap_selftest(uint32_t rounds, uint32_t lanes, uint64_t seed, int threads, uint64_t out[4]);
const char *oakgpu_last_error(void);
int main(void) {
  int bad = 0;
  for (int threads = 1; threads <= 16; threads *= 2)
    for (uint64_t seed = 1; seed <= 2; ++seed) {
      uint64_t out[4] = {0, 0, 0, 0};
      const int rc = oakgpu_heap_selftest(16, 2048, seed, threads, out);
      printf("threads %d seed %llu rc %d nodes %llu kept %llu end %llu violations %llu\n", threads, (unsigned long long)seed, rc,
             (unsigned long long)out[0], (unsigned long long)out[1], (unsigned long long)out[2], (unsigned long long)out[3]);
      if (rc || out[3]) { printf("  %s\n", oakgpu_last_error()); bad = 1; }
    }
  return bad;
}
"""


def test_threaded_tree_walk_is_clean_under_thread_sanitizer(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    lib = os.path.join(ROOT, "oak_amd", "liboakgpu.so")
    if not (os.path.exists(hipcc) and os.path.exists(CLANG) and os.path.exists(lib)):
        pytest.skip("needs hipcc, ROCm clang and the built library")
    obj, so, drv = str(tmp_path / "search_host_tsan.o"), str(tmp_path / "libsearch_tsan.so"), str(tmp_path / "drv")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O1", "-g", "-std=c++17", "-fPIC", "-Xarch_host", "-fsanitize=thread",
                           "-c", os.path.join(ROOT, "oak_amd", "csrc", "search_host.hip"), "-o", obj], stderr=subprocess.DEVNULL)
    # the sanitised search code first, everything else it calls from the product library
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-fsanitize=thread", "-o", so, obj,
                           "-L", os.path.dirname(lib), "-l:liboakgpu.so", "-Wl,-rpath," + os.path.dirname(lib)], stderr=subprocess.DEVNULL)
    (tmp_path / "drv.c").write_text(DRIVER)
    subprocess.check_call([CLANG, "-O1", "-g", "-fsanitize=thread", str(tmp_path / "drv.c"), "-o", drv, "-L", str(tmp_path), "-l:libsearch_tsan.so",
                           "-L", os.path.dirname(lib), "-l:liboakgpu.so", "-Wl,-rpath," + str(tmp_path), "-Wl,-rpath," + os.path.dirname(lib)])
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 exitcode=66")
    r = subprocess.run([drv], env=env, capture_output=True, text=True, timeout=600)
    out = r.stdout + r.stderr
    assert "ThreadSanitizer" not in out, out[-4000:]
    assert r.returncode == 0, out[-2000:]
    assert out.count("violations 0") == 10, out
