// Values the reference fixes in include/constants.h:3-18 and that are part of the contract
// (page geometry, sentinel token ids, segment order inside a page block).
#pragma once

constexpr int PAGE_BLOCK_SIZE = 16;          // tokens per KV page block
constexpr int DEFAULT_INIT_NUM_BLOCKS = 4;   // pages a row is admitted with
constexpr int EMPTY_ROW_TOKEN_ID = -1;       // decoder output for an empty slot
constexpr int EOF_TOKEN_ID = 1023;           // demo end-of-sequence token

constexpr int INP_EMB_EMB_OFFSET = 0;        // page segment order: input embedding, K, V
constexpr int K_CACHE_EMB_OFFSET = 1;
constexpr int V_CACHE_EMB_OFFSET = 2;

// Launch geometry constants of the reference (TILE_SIZE=16, WARP_SIZE=32, BLOCK_DIM=256) describe its
// CUDA kernels and have no meaning here: the HIP kernels choose their own tiling for 64-wide wavefronts.
constexpr int WAVE_SIZE = 64;
constexpr int BLOCK_DIM = 256;
