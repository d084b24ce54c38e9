// The Laplace SDF -> density conversion of VolSDF (code/src/engine/density.py:16-30) as the compositor evaluates it, in ONE
// place: the forward value (composite.hip, sampler.hip) and the exponential its backward multiplies every derivative by
// (composite.hip: d density / d sdf, d density / d beta).  hold_alive_* (points.hip) classifies a sample as DEAD when both are
// exact fp32 zeros -- then its compositing weight, the gradients of its colour and normal, d loss / d sdf and its share of
// d loss / d beta are all exactly zero and every per-sample stage behind the SDF can skip it without changing a bit of the
// result -- and must use the very same expressions.
#pragma once
#include <hip/hip_runtime.h>

__device__ __forceinline__ float hold_laplace_density(float s, float beta) {
  const float sg = (s > 0.f) ? 1.f : ((s < 0.f) ? -1.f : 0.f);
  return (1.0f / beta) * (0.5f + 0.5f * sg * expm1f(-fabsf(s) / beta));
}
// exp(-|s| / beta): the factor of d density / d sdf and of the s-dependent part of d density / d beta
__device__ __forceinline__ float hold_laplace_exp(float s, float beta) { return expf(-fabsf(s) / beta); }
