// ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see ora_common.h).
#ifndef ORA_KERNELS_H_
#define ORA_KERNELS_H_
#include "ora_core.h"

namespace ora {

float4 GenerateRandomNormal_YZL(Ctx& h, const Camera& camera, const int2 p, Rng& rng, const float depth);
void ComputeMultiViewCostVectorOld(const int2 p, float4 pl, float* cost_vector, Ctx& h);
void ComputeMultiViewCostVectorNew(const int2 p, float4 pl, float* cost_vector, Ctx& h);
void JointViewSelection(Ctx& h, int center, int iter, int phase, float cost_array[8][32],
	const float* view_selection_priors, uint8_t* view_weights, uint32_t* temp_selected_views, float* weight_norm);

// per-pixel kernel bodies
void RandomInitialization_px(Ctx& h, const int2 p);
void CheckerboardPropagationStrong_px(Ctx& h, const int2 p, const int iter);
void GetDepthandNormal_px(Ctx& h, const int2 p);
void CheckerboardFilterStrong_px(Ctx& h, const int2 p);
void DepthToWeak_px(Ctx& h, const int2 p);
void LocalRefine_px(Ctx& h, const int2 p);
void GenEdgeInform_px(Ctx& h, const int2 p);
void FindNearestStrongPoint_px(Ctx& h, const int2 p);
void GenNeighbours_px(Ctx& h, const int2 p);
void NeigbourUpdate_px(Ctx& h, const int2 p);
void RANSACToGetFitPlane_px(Ctx& h, const int2 p, int iter);
void CheckerboardPropagationWeak_px(Ctx& h, const int2 p, const int iter);

}  // namespace ora
#endif
