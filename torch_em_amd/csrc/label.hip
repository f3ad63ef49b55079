// label.hip -- training targets from instance labels, on device, integer compares only
// (bit-exact by construction).
//   tem_boundary_target: BoundaryTransform (reference transform/label.py:100-129) =
//     skimage.segmentation.find_boundaries(mode="thick"): grey-dilation != grey-erosion with
//     the 1-connectivity cross, i.e. "some face neighbour inside the volume differs".
//     mode="inner": thick & (label != 0).  mode="outer": thick & (background | adjacent), where adjacent =
//     foreground voxels whose full 3^ndim window has max(label) != min(label with background -> INT64_MAX),
//     i.e. two objects touch there (scikit-image segmentation/boundaries.py find_boundaries; scipy's reflect
//     border handling of the grey morphology == ignoring positions outside the volume).
//   tem_affinity_target: AffinityTransform (reference transform/label.py:248-327) =
//     1 - bioimage_cpp compute_affinities; semantics pinned by the reference's brute-force
//     definitions test/transform/test_label_transforms.py:5-55 (o = p + offset; outside the
//     volume -> aff 1, mask 0; ignore pairs -> aff 1, mask 0).
// HBM-bound streaming kernels; output is the reference's [C][D][H][W] float32 layout.
#include "tem_common.h"

__global__ __launch_bounds__(256) void k_boundary(const int64_t* __restrict__ lab, float* __restrict__ out, int D,
                                                  int H, int W, int add_binary, int mode) {
    const int64_t V = (int64_t)D * H * W;
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < V; v += (int64_t)gridDim.x * 256) {
        const int x = (int)(v % W);
        const int64_t r = v / W;
        const int y = (int)(r % H);
        const int z = (int)(r / H);
        const int64_t c = lab[v];
        bool b = false;
        if (x > 0) b |= lab[v - 1] != c;
        if (x < W - 1) b |= lab[v + 1] != c;
        if (y > 0) b |= lab[v - W] != c;
        if (y < H - 1) b |= lab[v + W] != c;
        if (z > 0) b |= lab[v - (int64_t)W * H] != c;
        if (z < D - 1) b |= lab[v + (int64_t)W * H] != c;
        if (mode == 1) {
            b = b && c != 0;
        } else if (mode == 2 && b && c != 0) {
            int64_t mx = c, mn = c;
            for (int dz = -1; dz <= 1; ++dz) {
                if (z + dz < 0 || z + dz >= D) continue;
                for (int dy = -1; dy <= 1; ++dy) {
                    if (y + dy < 0 || y + dy >= H) continue;
                    for (int dx = -1; dx <= 1; ++dx) {
                        if (x + dx < 0 || x + dx >= W) continue;
                        const int64_t q = lab[v + ((int64_t)dz * H + dy) * W + dx];
                        mx = q > mx ? q : mx;
                        const int64_t qi = q == 0 ? INT64_MAX : q;
                        mn = qi < mn ? qi : mn;
                    }
                }
            }
            b = mx != mn;
        }
        if (add_binary) {
            out[v] = (c != 0) ? 1.f : 0.f;
            out[V + v] = b ? 1.f : 0.f;
        } else {
            out[v] = b ? 1.f : 0.f;
        }
    }
}

extern "C" int tem_boundary_target_mode(const int64_t* labels, float* out, int D, int H, int W, int add_binary_target,
                                        int mode, tem_stream_t stream) {
    TEM_REQUIRE(labels && out && D > 0 && H > 0 && W > 0 && mode >= 0 && mode <= 2,
                "tem_boundary_target: bad arguments (mode: 0 thick, 1 inner, 2 outer)");
    int64_t V = (int64_t)D * H * W;
    hipLaunchKernelGGL(k_boundary, dim3(tem_grid_1d(V, 256)), dim3(256), 0, (hipStream_t)stream, labels, out, D, H, W,
                       add_binary_target, mode);
    TEM_CHECK_LAUNCH("tem_boundary_target");
    return TEM_OK;
}

extern "C" int tem_boundary_target(const int64_t* labels, float* out, int D, int H, int W, int add_binary_target,
                                   tem_stream_t stream) {
    return tem_boundary_target_mode(labels, out, D, H, W, add_binary_target, 0, stream);
}

#define AFF_MAX_OFF 64
struct AffOffsets {
    int o[AFF_MAX_OFF][3];
};

__global__ __launch_bounds__(256) void k_affinity(const int64_t* __restrict__ lab, float* __restrict__ out, int D,
                                                  int H, int W, AffOffsets offs, int n_off, int has_ignore,
                                                  int64_t ignore, int add_binary, int add_mask, int incl_trans) {
    const int64_t V = (int64_t)D * H * W;
    const int cb = add_binary ? 1 : 0;
    const int n_aff = n_off + cb;
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < V; v += (int64_t)gridDim.x * 256) {
        const int x = (int)(v % W);
        const int64_t r = v / W;
        const int y = (int)(r % H);
        const int z = (int)(r / H);
        const int64_t val = lab[v];
        if (add_binary) {
            out[v] = (val != 0) ? 1.f : 0.f;
            if (add_mask) out[(int64_t)n_aff * V + v] = (!has_ignore || val != ignore) ? 1.f : 0.f;
        }
        for (int c = 0; c < n_off; ++c) {
            const int oz = z + offs.o[c][0], oy = y + offs.o[c][1], ox = x + offs.o[c][2];
            float aff = 1.f, msk = 0.f;
            if (oz >= 0 && oz < D && oy >= 0 && oy < H && ox >= 0 && ox < W) {
                const int64_t oval = lab[((int64_t)oz * H + oy) * W + ox];
                int n_ign = 0;
                if (has_ignore) n_ign = (val == ignore) + (oval == ignore);
                if (n_ign == 2 || (n_ign == 1 && !incl_trans)) {
                    aff = 1.f;
                    msk = 0.f;
                } else {
                    aff = (val == oval) ? 0.f : 1.f;
                    msk = 1.f;
                }
            }
            out[(int64_t)(cb + c) * V + v] = aff;
            if (add_mask) out[(int64_t)(n_aff + cb + c) * V + v] = msk;
        }
    }
}

extern "C" int tem_affinity_target(const int64_t* labels, float* out, int D, int H, int W, const int* offsets,
                                   int n_off, int has_ignore, int64_t ignore_label, int add_binary_target, int add_mask,
                                   int include_ignore_transitions, tem_stream_t stream) {
    TEM_REQUIRE(labels && out && offsets && D > 0 && H > 0 && W > 0, "tem_affinity_target: bad arguments");
    TEM_REQUIRE(n_off > 0 && n_off <= AFF_MAX_OFF, "tem_affinity_target: between 1 and %d offsets supported", AFF_MAX_OFF);
    AffOffsets offs;
    for (int c = 0; c < n_off; ++c)
        for (int k = 0; k < 3; ++k) offs.o[c][k] = offsets[c * 3 + k];
    int64_t V = (int64_t)D * H * W;
    hipLaunchKernelGGL(k_affinity, dim3(tem_grid_1d(V, 256)), dim3(256), 0, (hipStream_t)stream, labels, out, D, H, W,
                       offs, n_off, has_ignore, ignore_label, add_binary_target, add_mask, include_ignore_transitions);
    TEM_CHECK_LAUNCH("tem_affinity_target");
    return TEM_OK;
}
