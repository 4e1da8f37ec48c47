// ngd_shard.h -- which shard computes which 128x128 pair tile (plain C++, no HIP: shared by the
// engine and by the host-side ngd_shard_of_pair()).
//
// Tiles of the upper triangle are enumerated row-major (ti <= tj).  An off-diagonal tile costs four
// full 64x64 jobs (4 x 16 MFMA tiles), a diagonal tile one full job and two triangular ones
// (16 + 2 x 10); tiles are dealt greedily, most expensive first, to the least loaded shard (ties: the
// lowest id).  Deterministic, so every rank derives the same map.
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

inline uint64_t ngd_tile_id(uint64_t n_t, uint64_t ti, uint64_t tj) {  // ti <= tj
  return ti * n_t - ti * (ti - 1) / 2 - ti + tj;
}

inline std::vector<uint32_t> ngd_tile_owners(uint32_t n_t, uint32_t world) {
  const uint64_t n_tiles = (uint64_t)n_t * (n_t + 1) / 2;
  std::vector<uint32_t> owner(n_tiles, 0);
  if (world <= 1) return owner;
  std::vector<uint64_t> load(world, 0);
  auto deal = [&](bool diag) {
    uint64_t id = 0;
    for (uint32_t ti = 0; ti < n_t; ti++)
      for (uint32_t tj = ti; tj < n_t; tj++, id++) {
        if ((ti == tj) != diag) continue;
        uint32_t best = 0;
        for (uint32_t r = 1; r < world; r++)
          if (load[r] < load[best]) best = r;
        owner[id] = best;
        load[best] += diag ? 36 : 64;
      }
  };
  deal(false);  // expensive tiles first
  deal(true);
  return owner;
}
