// Frames per second of the CMp3Enc-shaped one-stream API straight from C++ (no Python in the loop):
//   g++ -O2 -std=c++17 -Iinclude tools/bench_single.cpp -o /tmp/bench_single -Lhmp3_amd -lhmp3amd -Wl,-rpath,$PWD/hmp3_amd -Wl,-rpath,/opt/rocm/lib
//   /tmp/bench_single [frames]
// Replaces the loop of test/tomp3.cpp around CMp3Enc::L3_audio_encode (mp3enc.cpp:2031-2073).
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "hmp3_amd.h"

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 4000;
    HX_E_CONTROL ec;
    hx_default_control(&ec);
    ec.bitrate = 64;
    hx_enc *e = hx_enc_create(0);
    if (!e || !hx_enc_L3_audio_encode_init(e, &ec)) { fprintf(stderr, "init failed: %s\n", hx_last_error()); return 1; }
    std::vector<float> pcm(256 * 2304);
    unsigned r = 12345;
    for (size_t i = 0; i < pcm.size(); i++) {
        r = r * 1664525u + 1013904223u;
        pcm[i] = 6000.0f * sinf(0.013f * (float) (i / 2) * (1.0f + 0.3f * (float) (i & 1))) + (float) ((int) (r >> 20) - 2048);
    }
    std::vector<unsigned char> out(1 << 16);
    long long bytes = 0;
    for (int f = 0; f < 64; f++) hx_enc_L3_audio_encode(e, &pcm[(size_t) (f % 256) * 2304], out.data());
    const auto t0 = std::chrono::steady_clock::now();
    for (int f = 0; f < n; f++) bytes += hx_enc_L3_audio_encode(e, &pcm[(size_t) (f % 256) * 2304], out.data()).out_bytes;
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("hx_enc_L3_audio_encode (C++): %d frames in %.3f s = %.0f frames/s, %.1f us per call, %lld bytes\n", n, dt, n / dt, 1e6 * dt / n, bytes);
    hx_enc_destroy(e);
    return 0;
}
