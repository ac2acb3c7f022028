// Neighbour sets of the context-mapping weight generators, shared by the forward and backward kernels.
// VAR 0: eight_related (cmfsm.py:431-593): 9 planes c,l,r,t,b,lt,rt,lb,rb; tables 0,1,2,3,4,1,2,3,4
//   (5..8 alias 1..4, cmfsm.py:459-462, quirk Q2); out-of-image logit -100; output softmax.
// VAR 1: six_related, reference image (cmfsm_sub_8.py:449-572): 5 planes c,r,l,t,b with tables 0,1,2,3,4
//   (right uses table 1, left table 2: cmfsm_sub_8.py:503,525); out-of-image logit 0 (still in the softmax);
//   extra LeakyReLU after conv3 (:318,342); output softmax*logit (:572).
// VAR 2: six_related, target image: 3 planes c,r,l (cmfsm_sub_8.py:483-536), same MLP on the right-image features.
#pragma once
#include <hip/hip_runtime.h>

template <int VAR> struct Nbr;
template <> struct Nbr<0> {
    static constexpr int N = 9; static constexpr bool FINAL_ACT = false, TIMES_LOGIT = false;
    static constexpr float PAD = -100.f;
    static __device__ __forceinline__ int dy(int n) { constexpr int t[9] = {0, 0, 0, -1, 1, -1, -1, 1, 1}; return t[n]; }
    static __device__ __forceinline__ int dx(int n) { constexpr int t[9] = {0, -1, 1, 0, 0, -1, 1, -1, 1}; return t[n]; }
    static __device__ __forceinline__ int tab(int n) { constexpr int t[9] = {0, 1, 2, 3, 4, 1, 2, 3, 4}; return t[n]; }
};
template <> struct Nbr<1> {
    static constexpr int N = 5; static constexpr bool FINAL_ACT = true, TIMES_LOGIT = true;
    static constexpr float PAD = 0.f;
    static __device__ __forceinline__ int dy(int n) { constexpr int t[5] = {0, 0, 0, -1, 1}; return t[n]; }
    static __device__ __forceinline__ int dx(int n) { constexpr int t[5] = {0, 1, -1, 0, 0}; return t[n]; }
    static __device__ __forceinline__ int tab(int n) { constexpr int t[5] = {0, 1, 2, 3, 4}; return t[n]; }
};
template <> struct Nbr<2> {
    static constexpr int N = 3; static constexpr bool FINAL_ACT = true, TIMES_LOGIT = true;
    static constexpr float PAD = 0.f;
    static __device__ __forceinline__ int dy(int n) { return 0; }
    static __device__ __forceinline__ int dx(int n) { constexpr int t[3] = {0, 1, -1}; return t[n]; }
    static __device__ __forceinline__ int tab(int n) { constexpr int t[3] = {0, 1, 2}; return t[n]; }
};

__device__ __forceinline__ float ecm_centre_pat(int r, int s) { return (float)(r < s / 2 ? r - s / 2 : r - s / 2 + 1); }
// offset channel 0 (varies with X) / channel 1 (varies with Y) of table t at in-cell position r
__device__ __forceinline__ float ecm_off_x(int t, int r, int s) {
    return t == 1 ? (float)(s - r) : t == 2 ? (float)(r + 1) : ecm_centre_pat(r, s);
}
__device__ __forceinline__ float ecm_off_y(int t, int r, int s) {
    return t == 3 ? (float)(s - r) : t == 4 ? (float)(r + 1) : ecm_centre_pat(r, s);
}
