/*
 * lz4/lz4_index.hip.h -- LZ4 block format walker for the lane-per-chunk token indexer
 * (common/lz_index.hip.h). Finds where every sequence's token byte sits; the fields are
 * decoded later, lane-parallel, by lz4w::parse (which also does all the validation: this
 * walk only has to stay inside the chunk and terminate on any input).
 *
 * A step looks at the 8 stream bytes at the lane's position:
 *   mode 0  the token: literal length (with its first extension byte), then -- when the
 *           literals, the offset and the first match-length byte all sit inside the 8 bytes,
 *           the common case -- straight on to the next token;
 *   mode 1  further literal-length bytes (runs >= 270), 8 per step;
 *   mode 2  match-length bytes, 8 per step.
 */
#pragma once

#include "common/lz_index.hip.h"

namespace lz4i {

struct Format
{
  struct State
  {
    uint32_t mode;
    uint32_t acc;  /* mode 1: literal length so far */
    bool mext;     /* mode 1: the token's match length is extended */
  };

  static __device__ __forceinline__ void start(State& st)
  {
    st.mode = 0;
    st.acc = 0;
    st.mext = false;
  }

  /* Straight-line on the common path (all lanes at a token): every decision is a select; the length-byte modes
   * sit behind one wave-uniform test. */
  static __device__ __forceinline__ bool in_length_bytes(const State& st)
  {
    return st.mode != 0;
  }

  /* `any_slow`: some lane of the wave is at length bytes (wave-uniform; taken once per wave, outside divergent code) */
  static __device__ __forceinline__ bool step(State& st, uint32_t& p, uint64_t w, uint32_t vend, bool any_slow)
  {
    const uint32_t at = p;
    const uint32_t mode = st.mode;
    /* mode 0: token + first literal-length byte */
    const uint32_t t = (uint32_t)w & 0xffu;
    const uint32_t code = t >> 4;
    const uint32_t e1 = (uint32_t)(w >> 8) & 0xffu;
    const bool long_lit = code == 15 && e1 == 255;
    bool mext = (t & 15u) == 15u;
    uint32_t lit_end = code == 15 ? at + 17 + e1 : at + 1 + code;
    bool lit_done = !long_lit; /* the literal length is known: go on to offset / match length */
    uint32_t np = at + 2;      /* otherwise: next position / mode / accumulator */
    uint32_t nmode = 1;
    uint32_t nacc = 15 + 255;
    bool nmext = mext;
    if (any_slow) {
      const uint32_t k = lzi::first_not_255(w);
      const uint32_t b = (uint32_t)(w >> (8 * (k & 7u))) & 0xffu;
      const bool found = k < 8;
      if (mode != 0) {
        const uint32_t acc = st.acc + 255 * k + (found ? b : 0u);
        lit_done = mode == 1 && found;
        lit_end = at + k + 1 + acc;
        mext = st.mext;
        nmext = st.mext;
        nacc = acc;
        np = mode == 2 && found ? at + k + 1 : at + 8;
        nmode = found ? 0u : mode;
      }
    }
    /* literals end at lit_end: skip the offset and, when the first match-length byte is among the 8 bytes, that too */
    const uint32_t mpos = lit_end + 2;
    const uint32_t rel = mpos - at;
    const bool inwin = rel < 8;
    const uint32_t mb = (uint32_t)(w >> (8 * (rel & 7u))) & 0xffu;
    const bool ends = lit_end >= vend; /* last sequence (or a run past the end: the parser reports it) */
    const uint32_t p_seq = ends ? vend : !mext ? mpos : inwin ? mpos + 1 : mpos;
    const uint32_t m_seq = ends || !mext ? 0u : inwin ? (mb == 255 ? 2u : 0u) : 2u;
    p = lit_done ? p_seq : np;
    st.mode = lit_done ? m_seq : nmode;
    st.acc = nacc;
    st.mext = nmext;
    return mode == 0;
  }
};

} // namespace lz4i
