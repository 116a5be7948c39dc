// Host tessellator: Catmull-Clark limit-surface vertex grids for subdivision geometry.
//
// New code.  The reference evaluates each face through patch classification + bicubic B-spline patches +
// feature-adaptive subdivision with Gregory fill (kernels/subdiv/*, evalGrid subdivpatch1base_eval.cpp:77-215).
// All of these evaluate (up to the Gregory approximation in a 2^-10 neighbourhood of extraordinary vertices)
// the Catmull-Clark limit surface at the dyadic parameters (i/2^L, j/2^L).  This tessellator computes the same
// points directly: L rounds of uniform Catmull-Clark refinement of the whole control mesh in double precision
// followed by the limit-position (and, when a displacement callback is set, limit-tangent) stencils.  The
// fixed tessellation level 2^L per edge is the fork's (bvh_builder_subdiv.cpp:38-39,126-131,163-169).
#pragma once
#include "rt_objects.h"

namespace rtamd {

// One quad face tessellated to a (n+1) x (n+1) vertex grid, n = 2^L.  Row-major: index = j*(n+1)+i,
// i runs along face edge v0->v1 (parameter u), j along v0->v3 (parameter v).
struct PatchGrid
{
  unsigned geomID = 0, primID = 0;
  unsigned n = 0;
  std::vector<float> x, y, z;       // displaced positions (what evalGrid(..., applyDisplacement=true) returns)
  std::vector<float> bx, by, bz;    // undisplaced limit positions; empty when the geometry has no displacement
  // patch uv of grid point (i,j) is (u0 + i/n, v0 + j/n).  Quads: u0 = v0 = 0, uv corners (0,0),(1,0),(1,1),(0,1).
  // Sub-patch k of a non-quad face: (u0,v0) = (2*(k&3) + 0.5, 2*((k>>2)&3) + 0.5), the reference's encoding of the
  // sub-patch number in the integer part of uv (patch_eval_grid.h:241-254).
  float u0 = 0.f, v0 = 0.f;
  float px(size_t k, bool base) const { return base && !bx.empty() ? bx[k] : x[k]; }
  float py(size_t k, bool base) const { return base && !by.empty() ? by[k] : y[k]; }
  float pz(size_t k, bool base) const { return base && !bz.empty() ? bz[k] : z[k]; }
};

// Tessellate every valid face of a subdivision geometry at level L.  Calls the displacement callback on the host.
// `threads`: host threads for the chunked tessellation of large meshes (subdiv_tess.cpp tessellate_chunked).
void tessellate_subdiv(const Geometry* geom, unsigned geomID, unsigned L, std::vector<PatchGrid>& out, unsigned threads = 1);

// rtcInterpolate (row f4): triangle meshes (scene_triangle_mesh.cpp:214-270) and subdivision meshes (limit surface of any
// vertex / vertex-attribute buffer with first and second derivatives; scene_subdiv_mesh.cpp:757-864).
void interpolate_triangles(const Geometry* geom, const RTCInterpolateArguments* args);
void interpolate_subdiv(Geometry* geom, const RTCInterpolateArguments* args);

} // namespace rtamd
