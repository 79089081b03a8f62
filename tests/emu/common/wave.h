/*
 * tests/emu/common/wave.h -- TEST INFRASTRUCTURE ONLY.
 * CPU stand-in for nvcomp_amd/csrc/common/wave.h: same function set, each
 * cross-lane operation implemented as a rendezvous of the emulated wave's lanes
 * (see tests/emu/hip/hip_runtime.h).
 */
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace wave {

constexpr int kSize = 64;

struct u32x4
{
  uint32_t x, y, z, w;
};

enum OpId { kBallot = 1, kReadLane, kUniform, kShuffle, kScan, kMax, kSync, kLastWriter };

inline int lane_id() { return emu::cur()->lane; }
inline int fresh_lane_id() { return emu::cur()->lane; }
inline void touch(uint32_t&) {}
inline void sched_fence() {}
inline bool lane_in(uint64_t mask) { return ((mask >> emu::cur()->lane) & 1) != 0; }

inline uint64_t ballot(bool pred)
{
  emu::wave_rendezvous(kBallot, pred ? 1 : 0, 0);
  uint64_t m = 0;
  const uint64_t live = emu::live_mask();
  for (int i = 0; i < 64; ++i) {
    if (((live >> i) & 1) && emu::peer(i).a) {
      m |= 1ull << i;
    }
  }
  return m;
}

inline uint32_t read_lane(uint32_t v, uint32_t lane)
{
  emu::wave_rendezvous(kReadLane, v, lane);
  const uint64_t live = emu::live_mask();
  for (int i = 0; i < 64; ++i) {
    if (((live >> i) & 1) && emu::peer(i).b != lane) {
      fprintf(stderr, "emu: read_lane with a non-uniform lane index\n");
      abort();
    }
  }
  return (uint32_t)emu::peer((int)lane).a;
}

inline uint32_t uniform(uint32_t v)
{
  emu::wave_rendezvous(kUniform, v, 0);
  const uint64_t live = emu::live_mask();
  const int first = __builtin_ctzll(live);
  for (int i = 0; i < 64; ++i) {
    if (((live >> i) & 1) && emu::peer(i).a != emu::peer(first).a) {
      fprintf(stderr, "emu: uniform() of a value that differs between lanes\n");
      abort();
    }
  }
  return (uint32_t)emu::peer(first).a;
}

inline uint64_t uniform64(uint64_t v)
{
  emu::wave_rendezvous(kUniform, v, 0);
  const uint64_t live = emu::live_mask();
  const int first = __builtin_ctzll(live);
  for (int i = 0; i < 64; ++i) {
    if (((live >> i) & 1) && emu::peer(i).a != emu::peer(first).a) {
      fprintf(stderr, "emu: uniform64() of a value that differs between lanes\n");
      abort();
    }
  }
  return emu::peer(first).a;
}

template <typename T>
inline T* uniform_ptr(T* p)
{
  return (T*)uniform64((uint64_t)p);
}

template <typename T>
inline const T* kernel_args(const T& first_param)
{
  return &first_param;
}

inline uint32_t write_lane(uint32_t vec, uint32_t val, uint32_t lane)
{
  return ((uint32_t)lane_id() == lane) ? val : vec;
}

inline uint32_t write_lane_scalar(uint32_t vec, uint32_t val, uint32_t lane)
{
  return write_lane(vec, val, lane);
}

inline uint32_t read_lane(uint32_t v, uint32_t lane);
inline void chain_walk(uint32_t step, uint32_t limit, uint32_t& r, uint32_t& k, uint32_t& rec)
{
  do {
    const uint32_t d = read_lane(step, r);
    rec = write_lane(rec, r, k);
    ++k;
    r += d;
  } while (r < limit);
}

inline void select_walk(uint64_t candidates, uint32_t len, uint32_t start, uint64_t& taken, uint32_t& end)
{
  uint64_t r = candidates >> start;
  uint32_t pos = start;
  taken = 0;
  do {
    const uint32_t t = (uint32_t)__builtin_ctzll(r);
    pos += t;
    r >>= t;
    const uint32_t l = read_lane(len, pos);
    if (l >= 64) {
      fprintf(stderr, "emu: select_walk with a length of 64 or more\n");
      abort();
    }
    taken |= 1ull << pos;
    pos += l;
    r >>= l;
  } while (r);
  end = pos;
}

inline uint32_t gload_u8(const uint8_t* p) { return *p; }
inline uint32_t gload_u32(const uint8_t* p)
{
  uint32_t v;
  memcpy(&v, p, 4);
  return v;
}
inline u32x4 gload_u32x4_aligned(const uint8_t* p) { return *(const u32x4*)p; }
inline u32x4 gload_u32x4(const uint8_t* p)
{
  u32x4 v;
  memcpy(&v, p, 16);
  return v;
}
struct u32x3
{
  uint32_t x, y, z;
};
inline u32x3 gload_u32x3(const uint8_t* p)
{
  u32x3 v;
  memcpy(&v, p, 12);
  return v;
}
inline uint64_t gload_u64(const uint8_t* p)
{
  uint64_t v;
  memcpy(&v, p, 8);
  return v;
}
inline uint32_t gload_u16(const uint16_t* p) { return *p; }
inline void gstore_u8(uint8_t* p, uint32_t v) { *p = (uint8_t)v; }
inline void gstore_u32x4_aligned(uint8_t* p, u32x4 v) { *(u32x4*)p = v; }
inline void gstore_u32x4_aligned_nt(uint8_t* p, u32x4 v) { *(u32x4*)p = v; }
inline void gstore_u32(uint8_t* p, uint32_t v) { memcpy(p, &v, 4); }
inline void gstore_u32x4(uint8_t* p, u32x4 v) { memcpy(p, &v, 16); }

inline uint32_t shuffle(uint32_t v, uint32_t src_lane);
inline uint32_t prev_lane(uint32_t v)
{
  const uint32_t r = shuffle(v, (uint32_t)(lane_id() - 1) & 63u);
  return lane_id() == 0 ? 0u : r;
}

inline uint32_t shuffle(uint32_t v, uint32_t src_lane)
{
  emu::wave_rendezvous(kShuffle, v, src_lane);
  return (uint32_t)emu::peer((int)(src_lane & 63)).a;
}

inline uint32_t permute_to(uint32_t v, uint32_t dst_lane)
{
  emu::wave_rendezvous(kShuffle, v, dst_lane);
  uint32_t r = 0;
  int writers = 0;
  for (int i = 0; i < 64; ++i) {
    if (((uint32_t)emu::peer(i).b & 63u) == (uint32_t)lane_id()) {
      r = (uint32_t)emu::peer(i).a;
      ++writers;
    }
  }
  if (writers != 1) {
    fprintf(stderr, "emu: permute_to with destinations that are not a permutation\n");
    abort();
  }
  return r;
}

inline uint32_t next_lane(uint32_t v)
{
  const uint32_t r = shuffle(v, (uint32_t)(lane_id() + 1) & 63u);
  return lane_id() == 63 ? 0u : r;
}

inline uint32_t scan_add_inclusive(uint32_t v)
{
  emu::wave_rendezvous(kScan, v, 0);
  uint32_t s = 0;
  for (int i = 0; i <= lane_id(); ++i) {
    s += (uint32_t)emu::peer(i).a;
  }
  return s;
}

inline uint32_t umax(uint32_t a, uint32_t b) { return a > b ? a : b; }

inline uint32_t reduce_max(uint32_t v)
{
  emu::wave_rendezvous(kMax, v, 0);
  uint32_t m = 0;
  for (int i = 0; i < 64; ++i) {
    m = umax(m, (uint32_t)emu::peer(i).a);
  }
  return m;
}

inline uint32_t scan_max_inclusive(uint32_t v)
{
  emu::wave_rendezvous(kMax, v, 0);
  uint32_t m = 0;
  for (int i = 0; i <= lane_id(); ++i) {
    m = umax(m, (uint32_t)emu::peer(i).a);
  }
  return m;
}

inline void scan_last_writer(uint32_t& val, uint32_t& mask)
{
  emu::wave_rendezvous(kLastWriter, val & mask, mask);
  uint32_t v = 0, m = 0;
  for (int i = 0; i <= lane_id(); ++i) {
    const uint32_t pv = (uint32_t)emu::peer(i).a, pm = (uint32_t)emu::peer(i).b;
    v = (pv & pm) | (v & ~pm);
    m |= pm;
  }
  val = v;
  mask = m;
}

inline uint32_t reduce_add(uint32_t v)
{
  emu::wave_rendezvous(kScan, v, 0);
  uint32_t s = 0;
  for (int i = 0; i < 64; ++i) {
    s += (uint32_t)emu::peer(i).a;
  }
  return s;
}

inline void sync() { emu::wave_rendezvous(kSync, 0, 0); }
inline void sync_wave() { sync(); }

inline void lds_or(uint32_t* p, uint32_t bits) { *p |= bits; }
inline void lds_add(uint32_t* p, uint32_t v) { *p += v; }
inline void lds_sub(uint32_t* p, uint32_t v) { *p -= v; }
inline void lds_store_release(uint32_t* p, uint32_t v) { *(volatile uint32_t*)p = v; }
inline uint32_t lds_load_acquire(const uint32_t* p) { return *(const volatile uint32_t*)p; }
inline void lds_store_relaxed(uint32_t* p, uint32_t v) { *(volatile uint32_t*)p = v; }
inline uint32_t lds_load_relaxed(const uint32_t* p) { return *(const volatile uint32_t*)p; }
inline void nap() { sync(); }
inline void nap_short() { sync(); } /* a rendezvous is where the coroutine scheduler lets the other wave run */

inline uint32_t ctz64(uint64_t m) { return (uint32_t)__builtin_ctzll(m); }
inline uint32_t popc64(uint64_t m) { return (uint32_t)__builtin_popcountll(m); }
inline uint32_t mul24(uint32_t a, uint32_t b) { return (a & 0xffffffu) * (b & 0xffffffu); }
inline uint32_t align_bits(uint32_t hi, uint32_t lo, uint32_t shift)
{
  return (uint32_t)((((uint64_t)hi << 32) | lo) >> (shift & 31u));
}
inline uint32_t align_bytes(uint32_t hi, uint32_t lo, uint32_t shift)
{
  return (uint32_t)((((uint64_t)hi << 32) | lo) >> (8 * (shift & 3u)));
}
inline uint32_t pk_add_sat255(uint32_t a, uint32_t b)
{
  uint32_t lo = (a & 0xffffu) + (b & 0xffffu), hi = (a >> 16) + (b >> 16);
  lo &= 0xffffu;
  hi &= 0xffffu;
  lo = lo < 255u ? lo : 255u;
  hi = hi < 255u ? hi : 255u;
  return lo | (hi << 16);
}
inline uint32_t bit_reverse(uint32_t v)
{
  v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
  v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
  v = ((v >> 4) & 0x0f0f0f0fu) | ((v & 0x0f0f0f0fu) << 4);
  return __builtin_bswap32(v);
}

inline uint32_t perm_bytes(uint32_t hi, uint32_t lo, uint32_t sel)
{
  const uint64_t v = ((uint64_t)hi << 32) | lo;
  uint32_t r = 0;
  for (int i = 0; i < 4; ++i) {
    const uint32_t k = (sel >> (8 * i)) & 0xffu;
    const uint32_t b = k < 8 ? (uint32_t)(v >> (8 * k)) & 0xffu : k == 12 ? 0u : 0xffu;
    r |= b << (8 * i);
  }
  return r;
}
inline uint32_t prefix_popc(uint64_t m)
{
  const uint32_t l = (uint32_t)lane_id();
  return (uint32_t)__builtin_popcountll(m & ((1ull << l) - 1ull));
}

} // namespace wave

/* statistics hook: lane 0 of a wave accumulates named counters (see emu_stats_dump) */
extern "C" void emu_stat_add(const char* name, unsigned long long n);
#define WAVE_DYNAMIC_LDS(name) uint8_t* name = emu::dynamic_lds()
#define LZ_STAT(name, n)                  \
  do {                                    \
    const unsigned long long v_ = (n);    \
    if (emu::cur()->lane == 0) {          \
      emu_stat_add(name, v_);             \
    }                                     \
  } while (0)
