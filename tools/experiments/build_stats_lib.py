"""Builds tools/experiments/libgsr_stats.so: the library with iteration counters spliced into a copy of
the blend kernels (the shipped sources stay free of instrumentation).  Counters, read with
gsr_blend_stats(out[8], reset): forward [0] wave iterations started, [1] voted in, [2] lanes taking part,
[3] splats staged; backward [4..7] likewise ([7] staged)."""
import os
import shutil
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))


def splice(src: str) -> str:
    def once(s, old, new):
        assert s.count(old) == 1, old
        return s.replace(old, new)
    s = once(src, "namespace {\n\nconstexpr int kBatch = 256;", '''__device__ unsigned long long g_stats[8];
extern "C" int gsr_blend_stats(unsigned long long* out, int reset) {
  if (reset) { unsigned long long z[8] = {0}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_stats), z, sizeof(z)); }
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stats), 64);
}
namespace {

constexpr int kBatch = 256;''')
    s = once(s, "          const bool keep = (p <= 0.0f) & (oG >= accept);\n", '''          const bool keep = (p <= 0.0f) & (oG >= accept);
          { const unsigned long long bk = __ballot(keep);
            if (lane == 0) { atomicAdd(&g_stats[0], 1ull); if (bk) { atomicAdd(&g_stats[1], 1ull); atomicAdd(&g_stats[2], (unsigned long long)__popcll(bk)); } } }
''')
    s = once(s, "        if (__ballot(hit) == 0ull) continue;", '''        { const unsigned long long bk = __ballot(hit);
          if (lane == 0) { atomicAdd(&g_stats[4], 1ull); if (bk) { atomicAdd(&g_stats[5], 1ull); atomicAdd(&g_stats[6], (unsigned long long)__popcll(bk)); } } }
        if (__ballot(hit) == 0ull) continue;''')
    staged = "      stage_record(&s_splat[threadIdx.x], r0, r1, r2, g);\n"
    assert s.count(staged) == 2                      # forward first, backward second
    first = s.index(staged)
    second = s.index(staged, first + 1)
    s = s[:second] + staged + "      atomicAdd(&g_stats[7], 1ull);\n" + s[second + len(staged):]
    s = s[:first] + staged + "      atomicAdd(&g_stats[3], 1ull);\n" + s[first + len(staged):]
    return s


def main():
    root = os.path.abspath(os.path.join(HERE, "..", ".."))
    tmp = tempfile.mkdtemp()
    try:
        shutil.copytree(os.path.join(root, "include"), os.path.join(tmp, "include"))
        csrc = os.path.join(tmp, "pkg", "csrc")
        shutil.copytree(os.path.join(root, "gaussian-splatting-own_amd", "csrc"), csrc,
                        ignore=shutil.ignore_patterns("_obj"))
        os.makedirs(os.path.join(tmp, "pkg", "diff_gaussian_rasterization"))
        with open(os.path.join(csrc, "blend.hip")) as fh:
            src = fh.read()
        with open(os.path.join(csrc, "blend.hip"), "w") as fh:
            fh.write(splice(src))
        subprocess.check_call(["make", "-C", csrc, "-j4"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        shutil.copy(os.path.join(tmp, "pkg", "diff_gaussian_rasterization", "libgsr_hip.so"),
                    os.path.join(HERE, "libgsr_stats.so"))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
