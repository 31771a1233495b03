// CPU-side sanitizer run of the library's HOST code (race / memory checking; GPU AddressSanitizer is not
// available on the pool, so the host half is checked on the CPU build): every source of music_amd/csrc is compiled
// --offload-host-only with -fsanitize=address,undefined and this driver walks the entry points that do host work only -
// the launch plans (item walks of the chain form, slab counts, hand-off area sizes) over many shapes, and the argument
// checks of the compute entry points (they must return their status before anything touches a device).
// Built and run by tests/test_host_sanitize.py; exit code 0 and no sanitizer report = pass.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../include/wavenet_hip.h"

static int fails = 0;
#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "host_sanitize: %s failed (line %d)\n", #c, __LINE__); ++fails; } } while (0)

int main() {
    CHECK(wn_version() == WN_ABI_VERSION);
    // ---- chain plan: every item of every clip owned exactly once, over a grid of shapes (the python test pins a few by hand)
    long walked = 0;
    for (int d = 32; d <= 512; d *= 2)
        for (int batch : {1, 2, 3, 8, 33, 64})
            for (int t_lo : {1, 40, 511, 1024, 3071})
                for (int len : {d, d + 1, 3 * d + 17, 5000, 13000}) {
                    const int t_hi = t_lo + len;
                    if (!wn_resblock_bwd_pq_chain_ok(t_lo, t_hi, batch, d)) continue;
                    const int nwg = wn_resblock_bwd_pq_slabs(t_lo, t_hi, batch, d, 1);
                    CHECK(nwg >= 1 && nwg <= 256);
                    const int t_base = t_lo & ~31, steps = (t_hi - t_base + 31) / 32;
                    std::vector<unsigned char> seen((size_t)batch * steps, 0);
                    std::vector<int> out(3 * 8192);
                    for (int wg = 0; wg < nwg; ++wg) {
                        const int n = wn_resblock_bwd_pq_chain_items(t_lo, t_hi, batch, d, wg, out.data(), 8192);
                        CHECK(n >= 0 && n <= 8192);
                        for (int k = 0; k < n; ++k) {
                            const int b = out[3 * k], t0 = out[3 * k + 1], fl = out[3 * k + 2];
                            CHECK(b >= 0 && b < batch && t0 >= t_base && (t0 - t_base) % 32 == 0 && (t0 - t_base) / 32 < steps);
                            if (fl & 1) { CHECK(k == 0); continue; }               // halo item: owned by the workgroup above
                            unsigned char& s = seen[(size_t)b * steps + (t0 - t_base) / 32];
                            CHECK(s == 0);
                            s = 1;
                            ++walked;
                        }
                    }
                    for (unsigned char s : seen) CHECK(s == 1);
                    CHECK(wn_resblock_bwd_pq_chain_items(t_lo, t_hi, batch, d, nwg, out.data(), 8192) == -1);
                }
    CHECK(walked > 100000);
    CHECK(wn_resblock_bwd_pq_chain_ok(100, 16000, 8, 16) == 0 && wn_resblock_bwd_pq_chain_ok(100, 16000, 8, 48) == 0);
    // ---- slab counts / hand-off sizes: positive, monotone in the work
    for (int batch : {1, 8, 64})
        for (int t_hi : {400, 4000, 16000, 160000}) {
            const int a = wn_resblock_bwd_ms_slabs(100, t_hi, batch), b = wn_wgrad_slabs(100, t_hi, 512, batch);
            CHECK(a >= 1 && a <= 256 && b >= batch);
            CHECK(wn_resblock_bwd_pq_slabs(100, t_hi, batch, 4, 0) == a);
            CHECK(wn_enc_resblock_bwd_slabs(100, t_hi, batch) >= 1 && wn_causal_wgrad_codes_slabs(t_hi, batch) >= 1);
            CHECK(wn_resblock_bwd_pq_cond_floats(100, t_hi, batch) > 0);
        }
    CHECK(wn_decode_sync_granules(30, 64, 256) == 30 * 64 + 2 * 256 + 256 + 2);
    CHECK(wn_decode_sync_granules(40, 64, 512) == 40 * 64 + 2 * 512 + 256 + 40 * 512 + 2);
    // ---- argument checks: a status and a message, nothing launched (no device is needed for any of these)
    float buf[64] = {0};
    uint16_t pk[64] = {0};
    int32_t codes[8] = {0};
    CHECK(wn_resblock_bwd_pq(buf, nullptr, nullptr, 0, 0, buf, buf, buf, 64, 64, 6, pk, pk, pk, 64, 1, 8, 200, 8, buf, buf, nullptr, 0, 0, 0,
                             nullptr, nullptr, 0, 0, 1, 0, 2, nullptr) != 0);                     // pitch % 4
    CHECK(std::strlen(wn_last_error()) > 0);
    CHECK(wn_resblock_bwd_pq(buf, nullptr, nullptr, 0, 0, buf, buf, buf, 64, 64, 8, pk, pk, pk, 32, 1, 8, 200, 8, buf, buf, nullptr, 0, 0, 0,
                             nullptr, nullptr, 0, 0, 1, 0, 2, nullptr) != 0);                     // 32 channels
    CHECK(wn_resblock_bwd_pq(buf, nullptr, buf, 0, 0, buf, buf, buf, 64, 64, 8, pk, pk, pk, 64, 1, 8, 200, 8, buf, buf, nullptr, 0, 0, 0,
                             nullptr, nullptr, 0, 0, 1, 0, 2, nullptr) != 0);                     // q_in without p_in
    CHECK(wn_resblock_bwd_pq(buf, nullptr, nullptr, 0, 0, buf, buf, buf, 64, 64, 8, pk, pk, pk, 64, 1, 8, 200, 8, buf, buf, buf, 0, 0, 40,
                             nullptr, nullptr, 0, 0, 1, 0, 2, nullptr) != 0);                     // conditioned: no bucket bytes
    CHECK(wn_mulaw_encode_q(buf, buf, 1, codes, 8, nullptr) != 0);
    CHECK(wn_mulaw_decode_q(codes, buf, 0, buf, 8, nullptr) != 0);
    CHECK(wn_chunk_softmax_fwd(buf, buf, 4, 0, nullptr) != 0);
    CHECK(wn_split16(nullptr, nullptr, nullptr, 8, 1, nullptr) != 0);
    CHECK(wn_gather_grads2(nullptr, nullptr, nullptr, nullptr, 8, nullptr) != 0);
    if (fails) { std::fprintf(stderr, "host_sanitize: %d check(s) failed\n", fails); return 1; }
    std::printf("host_sanitize ok: %ld chain items walked\n", walked);
    return 0;
}
