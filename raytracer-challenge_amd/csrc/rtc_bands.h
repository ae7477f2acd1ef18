// rtc_bands.h — the ONE definition of how the rows of a frame are dealt over the N members of an rtc_group
// (include/rtc.h): the canvas (canvas.rs:43-51, row-major) is cut into bands of RTC_BAND_ROWS = 8 rows (the render
// kernel's tile height), band b belongs to member b % N and is the (b / N)-th band of that member's PACKED tile. Used by
//   * rtc_group.cpp   — tile sizes, the gather's chunk layout, the host-canvas DMA offsets,
//   * k_undeal        — staging chunk -> row-major canvas on member 0 (rtc_kernels.hip),
//   * the [host] entries rtc_group_packed_rows / _bands_owned / _row_owner / _packed_row_to_image / _undeal_host
//     (include/rtc.h), through which tests/test_bands_gloo.py checks THIS arithmetic with two real ranks on the CPU.
// (The render kernel's own row selection — tile row k of a launch renders image rows y0 + 8*k*band_stride — is the same
// rule with y0 = 8*rank, band_stride = N; it is pinned against single-GPU renders by tests/test_gpu_group.py.)
#ifndef RTC_BANDS_H
#define RTC_BANDS_H

#include <stddef.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define RTC_HD __host__ __device__ inline
#else
#define RTC_HD inline
#endif

#define RTC_BANDS_ROWS 8u // == RTC_BAND_ROWS (include/rtc.h), static_assert'ed in rtc_group.cpp

// bands of a frame of `vsize` rows (the last one possibly short)
RTC_HD uint32_t rtc_bands_of(uint32_t vsize) { return (vsize + RTC_BANDS_ROWS - 1u) / RTC_BANDS_ROWS; }
// bands member `rank` of `nranks` owns: rank, rank + nranks, ...
RTC_HD uint32_t rtc_bands_owned(uint32_t vsize, uint32_t nranks, uint32_t rank) {
    const uint32_t nb = rtc_bands_of(vsize);
    return rank >= nb ? 0u : (nb - rank + nranks - 1u) / nranks;
}
// rows of one member's packed tile: the most bands any member owns (member 0), 8 rows each — the same for every
// member, so that the gather's chunks are equal
RTC_HD uint32_t rtc_packed_rows(uint32_t vsize, uint32_t nranks) { return ((rtc_bands_of(vsize) + nranks - 1u) / nranks) * RTC_BANDS_ROWS; }
// image row y -> owning member and row inside that member's packed tile
RTC_HD void rtc_row_owner(uint32_t y, uint32_t nranks, uint32_t *member, uint32_t *packed_row) {
    const uint32_t band = y / RTC_BANDS_ROWS;
    *member = band % nranks;
    *packed_row = (band / nranks) * RTC_BANDS_ROWS + (y % RTC_BANDS_ROWS);
}
// ... and back: row r of member p's packed tile -> image row (may be >= vsize: padding of the last band / of members
// that own one band less)
RTC_HD uint32_t rtc_packed_row_to_image(uint32_t member, uint32_t packed_row, uint32_t nranks) {
    return (member + (packed_row / RTC_BANDS_ROWS) * nranks) * RTC_BANDS_ROWS + (packed_row % RTC_BANDS_ROWS);
}
// The gather leaves N chunks in member 0's staging buffer, chunk p = member p's packed tiles of `nframes` frames
// ([nframes][rows_max][row_units] units). Index of unit u of image row y of frame f:
RTC_HD size_t rtc_staging_index(uint32_t f, uint32_t y, uint32_t u, uint32_t nranks, uint32_t nframes, uint32_t rows_max, uint32_t row_units) {
    uint32_t p, r;
    rtc_row_owner(y, nranks, &p, &r);
    return (((size_t)p * nframes + f) * rows_max + r) * row_units + u;
}

#endif
