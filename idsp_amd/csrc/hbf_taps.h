// hbf_taps.h — the reference's built-in half-band tap sets as compile-time
// tables: HBF_TAPS (140 dB, src/hbf.rs:308-349) and HBF_TAPS_98 (src/hbf.rs:258-292).
// Second index = the reference's tuple index (0 = lowest-rate stage); taps run
// from the outermost (small) to the centre (large) tap.
#pragma once

namespace idsp {

constexpr int kHbfM[2][5] = {{23, 10, 5, 4, 3}, {15, 6, 3, 3, 2}};
constexpr float kHbfTaps[2][5][23] = {
    {{7.60375795e-07f, -3.77494111e-06f, 1.26458559e-05f, -3.43188253e-05f, 8.10687478e-05f, -1.72971467e-04f,
      3.40845059e-04f, -6.29522864e-04f, 1.10128831e-03f, -1.83933299e-03f, 2.95124926e-03f, -4.57290964e-03f,
      6.87374176e-03f, -1.00656257e-02f, 1.44199840e-02f, -2.03025100e-02f, 2.82462332e-02f, -3.91128509e-02f,
      5.44795658e-02f, -7.77002672e-02f, 1.17523452e-01f, -2.06185388e-01f, 6.34588695e-01f},
     {-1.12811343e-05f, 1.12724671e-04f, -6.07439343e-04f, 2.31904511e-03f, -7.00322950e-03f, 1.78225473e-02f,
      -4.01209836e-02f, 8.43315989e-02f, -1.83189521e-01f, 6.26346521e-01f},
     {0.0007686f, -0.00768669f, 0.0386536f, -0.14002434f, 0.60828885f},
     {-0.00261331f, 0.02476858f, -0.12112638f, 0.59897111f},
     {0.01186105f, -0.09808109f, 0.58622005f}},
    {{7.02144012e-05f, -2.43279582e-04f, 6.35026936e-04f, -1.39782541e-03f, 2.74613582e-03f, -4.96403839e-03f,
      8.41806912e-03f, -1.35827601e-02f, 2.11004053e-02f, -3.19267647e-02f, 4.77024289e-02f, -7.18014345e-02f,
      1.12942004e-01f, -2.03279594e-01f, 6.33592923e-01f},
     {-0.00086943f, 0.00577837f, -0.02201674f, 0.06357869f, -0.16627679f, 0.61979312f},
     {0.01414651f, -0.10439639f, 0.59026742f},
     {0.01227974f, -0.09930782f, 0.58702834f},
     {-0.06291796f, 0.5629161f}},
};


// Tuple index of processing stage s in a 2^stages cascade: the decimator runs the
// highest-rate stage first (HBF_DEC_CASCADE, src/hbf.rs:412-421), the
// interpolator the lowest-rate stage first (HBF_INT_CASCADE, src/hbf.rs:503-512).
constexpr int hbf_tuple_index(bool dec, int stages, int s) { return dec ? stages - 1 - s : s; }

}  // namespace idsp
