// evc_bits.h - bit-level layer of the host front end: bit writer / reader, the arithmetic decoder and its mirror-image encoder, the binarisations both share,
// NAL unit framing.  Included by evc_parser.cc and evc_writer.cc (everything lives in an anonymous namespace: each translation unit has its own copy).
#pragma once
#include "../../include/xevd_host.h"
#include <algorithm>
#include <atomic>
#include <map>
#include <memory>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

// ------------------------------------------------------------------------------------------------ bit I/O
struct BitWriter {
    std::vector<uint8_t> buf;
    uint32_t acc = 0;
    int n = 0;
    void put1(int b) { acc = (acc << 1) | (uint32_t)(b & 1); if (++n == 8) { buf.push_back((uint8_t)acc); acc = 0; n = 0; } }
    void put(uint32_t v, int len) { for (int i = len - 1; i >= 0; i--) put1((int)((v >> i) & 1)); }
    void ue(uint32_t v) { const uint64_t x = (uint64_t)v + 1; int len = 0; while ((x >> len) > 1) len++; for (int i = 0; i < len; i++) put1(0); for (int i = len; i >= 0; i--) put1((int)((x >> i) & 1)); }
    void se(int v) { ue(v > 0 ? (uint32_t)(2 * v - 1) : (uint32_t)(-2 * v)); }      // xevd_bsr_read_se, xevd_bsr.c:322-328
    void align_zero() { while (n) put1(0); }
};
struct BitReader {
    const uint8_t *p = nullptr;
    size_t size = 0, pos = 0;       // pos in bits
    bool overrun = false;
    int get1() { if (pos >= size * 8) { overrun = true; return 0; } const int b = (p[pos >> 3] >> (7 - (pos & 7))) & 1; pos++; return b; }
    uint32_t get(int len) { uint32_t v = 0; for (int i = 0; i < len; i++) v = (v << 1) | (uint32_t)get1(); return v; }
    uint32_t ue() { int z = 0; while (!get1()) { if (++z > 32 || overrun) { overrun = true; return 0; } } uint64_t v = 1; for (int i = 0; i < z; i++) v = (v << 1) | (uint64_t)get1(); return (uint32_t)(v - 1); }
    int se() { const uint32_t k = ue(); return (k & 1) ? (int)((k + 1) >> 1) : -(int)(k >> 1); }
    bool aligned() const { return (pos & 7) == 0; }
};

// ------------------------------------------------------------------------------------------------ arithmetic coder
// Context model = (state << 1) | mps, state 9 bits, initial 512 = state 256 (p = 1/2) - xevd_eco.c:35-87, xevd_def.h:76
typedef uint16_t Model;
static inline void model_update(Model &m, bool lps)
{
    int state = m >> 1, mps = m & 1;
    if (lps) { state = state + ((512 - state + 16) >> 5); if (state > 256) { mps = 1 - mps; state = 512 - state; } }
    else state = state - ((state + 16) >> 5);
    m = (Model)((state << 1) | mps);
}
static inline uint32_t lps_range(Model m, uint32_t range) { const uint32_t l = ((uint32_t)(m >> 1) * range) >> 9; return l < 437 ? 437 : l; }

struct Dec {      // xevd_sbac_decode_bin / sbac_decode_bin_ep / xevd_sbac_decode_bin_trm, xevd_eco.c:35-165
    BitReader *br;
    uint32_t range, value;
    // The bits of the tile come through a 64-bit window (the reference shifts the stream in bit by bit, and so did round 1-3's get1() per renormalisation step: a
    // bounds check, a byte index and a shift for every bit): refilled a byte at a time, zeros past the end of the data - reading one of THOSE sets br->overrun, as
    // BitReader::get1 does.  br->pos is brought up to date by sync() (tile_end, and whoever looks at the reader after the tile).
    uint64_t win = 0;
    int avail = 0;                   // unread bits in `win`
    size_t byte_pos = 0;             // next byte of br->p to load
    void refill()
    {
        while (avail <= 56) { win = (win << 8) | (uint64_t)(byte_pos < br->size ? br->p[byte_pos] : 0); byte_pos++; avail += 8; }
    }
    uint32_t take(int n)             // 1 <= n <= 16 bits, first bit in the most significant place
    {
        if (avail < n) refill();
        avail -= n;
        if (byte_pos > br->size && byte_pos * 8 - (size_t)avail > br->size * 8) br->overrun = true;
        return (uint32_t)(win >> avail) & ((1u << n) - 1u);
    }
    void sync() { br->pos = byte_pos * 8 - (size_t)avail; }
    void start()
    {
        byte_pos = br->pos >> 3; win = 0; avail = 0;
        refill();
        avail -= (int)(br->pos & 7);                                 // a tile starts at a byte boundary in every stream; kept general
        range = 16384; value = take(14);
    }
    int bin(int, Model &m)
    {
        const int mps = m & 1;
        const uint32_t lps = lps_range(m, range);
        int b = mps;
        range -= lps;
        if (value >= range) { b = 1 - mps; value -= range; range = lps; model_update(m, true); }
        else model_update(m, false);
        if (range < 8192) {                                          // renormalise: all the missing bits at once
            const int sh = __builtin_clz(range) - 18;                // range < 2^13: bring its top bit to bit 13
            range <<= sh;
            value = ((value << sh) | take(sh)) & 0xFFFF;
        }
        return b;
    }
    int ep(int)
    {
        int b = 0;
        range >>= 1;
        if (value >= range) { b = 1; value -= range; }
        range <<= 1;
        value = ((value << 1) | take(1)) & 0xFFFF;
        return b;
    }
    int tile_end()      // terminating bin, then zero bits up to the byte boundary and zero words up to the end (xevd_eco.c:100-140,1683-1695)
    {
        range--;
        sync();
        if (value < range) return 0;
        while (!br->aligned()) if (br->get1()) return -1;
        return 1;
    }
};
struct Enc {      // the mirror image: MPS takes the lower part of the interval, carries resolved with outstanding bits
    BitWriter *bw;
    uint32_t low = 0, range = 16384;
    int outstanding = 0;
    bool first = true;
    void start() { low = 0; range = 16384; outstanding = 0; first = true; }
    void emit(int b) { if (first) first = false; else bw->put1(b); while (outstanding) { bw->put1(!b); outstanding--; } }
    void shift_out()
    {
        if (low < 8192) emit(0);
        else if (low >= 16384) { low -= 16384; emit(1); }
        else { low -= 8192; outstanding++; }
        low <<= 1;
    }
    int bin(int v, Model &m)
    {
        const int mps = m & 1;
        const uint32_t lps = lps_range(m, range);
        range -= lps;
        if ((v & 1) != mps) { low += range; range = lps; model_update(m, true); }
        else model_update(m, false);
        while (range < 8192) { shift_out(); range <<= 1; }
        return v & 1;
    }
    int ep(int v)
    {
        const uint32_t half = range >> 1;
        if (v & 1) low += half;
        shift_out();
        range = half << 1;
        return v & 1;
    }
    int tile_end()
    {
        range--;
        low += range;                               // the top unit of the interval: the decoder sees value >= range
        emit((int)((low >> 14) & 1));
        for (int i = 13; i >= 0; i--) bw->put1((int)((low >> i) & 1));
        bw->align_zero();
        return 1;
    }
};

// Symbol binarisations shared by both coders (xevd_eco.c:167-258, 452-489)
template <class C> static int sym_unary(C &c, int v, Model *m, int num_ctx)       // sbac_read_unary_sym
{
    int sym = 0, ctx = 0;
    if (!c.bin(v > 0, m[0])) return 0;
    for (;;) {
        if (ctx < num_ctx - 1) ctx++;
        sym++;
        if (!c.bin(v > sym, m[ctx])) break;
        if (sym >= 0x7FFF) break;                   // no valid symbol is longer (levels are s16, runs below 4096): malformed input ends here
    }
    return sym;
}
template <class C> static int sym_trunc_unary(C &c, int v, Model *m, int num_ctx, int max_num)      // sbac_read_truncate_unary_sym
{
    int i = 0;
    if (max_num > 1)
        for (; i < max_num - 1; ++i)
            if (!c.bin(v > i, m[i > num_ctx - 1 ? num_ctx - 1 : i])) break;
    return i;
}
template <class C> static int sym_abs_mvd(C &c, int v, Model &m)      // xevd_eco_abs_mvd: 1 = zero; else (len-1) zeros + 1, then len suffix bits
{
    if (c.bin(v == 0, m)) return 0;
    int len_v = 0;
    while (((v + 1) >> (len_v + 1)) > 0) len_v++;            // floor(log2(v + 1)) on the encoder side
    int len = 0, code;
    do { code = len == 0 ? c.bin(len + 1 == len_v, m) : c.ep(len + 1 == len_v); len++; } while (!code && len < 24);      // a valid |mvd| has at most 16 prefix bins
    int val = (1 << len) - 1;
    const int suffix = v + 1 - (1 << len_v);
    while (len != 0) { len--; val += c.ep((suffix >> len) & 1) << len; }
    return val;
}

template <class C> static int sym_unary_ep(C &c, int v, int max_val)                 // sbac_read_unary_sym_ep, xevd_eco.c:166-189
{
    if (!c.ep(v > 0)) return 0;
    int sym = 0, counter = 1, t;
    do { t = counter == max_val ? 0 : c.ep(v > sym + 1); counter++; sym++; } while (t);
    return sym;
}
template <class C> static int sym_bits_ep(C &c, int v, int n)                         // sbac_decode_bins_ep: most significant bin first
{
    int r = 0;
    for (int i = n - 1; i >= 0; i--) r = (r << 1) | c.ep((v >> i) & 1);
    return r;
}


enum { NUT_NONIDR = 0, NUT_IDR = 1, NUT_SPS = 24, NUT_PPS = 25, NUT_SEI = 28 };

static inline void write_nal(std::vector<uint8_t> &out, int nut, int tid, const BitWriter &payload)
{
    const uint32_t len = (uint32_t)payload.buf.size() + 2;
    for (int i = 3; i >= 0; i--) out.push_back((uint8_t)(len >> (8 * i)));
    const uint32_t hdr = ((uint32_t)(nut + 1) << 9) | ((uint32_t)tid << 6);      // 1 zero bit, type + 1 (6), tid (3), 5 reserved zero bits, 1 extension bit
    out.push_back((uint8_t)(hdr >> 8)); out.push_back((uint8_t)hdr);
    out.insert(out.end(), payload.buf.begin(), payload.buf.end());
}

}   // namespace
