// urt_types.h — byte layouts the C-ABI accepts.  They are the reference's own structured-buffer
// layouts (C# sequential structs, 4-byte fields, no padding):
//   RayTraceParams  40 B   Assets/Scripts/RayTraceMaster.cs:48-53  / RayTraceShader.compute:29-34
//   MeshObject     112 B   RayTraceMaster.cs:82-86                 / RayTraceShader.compute:43-49
//   Sphere          56 B   RayTraceMaster.cs:116-119               / RayTraceShader.compute:51-55
//   BVHNode         28 B   RayTraceMaster.cs:148-152               / RayTraceShader.compute:57-61
// Strides are asserted by the reference at RayTraceMaster.cs:42-45 and used at :738-745.
#pragma once
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#pragma pack(push, 1)
typedef struct urt_RayTraceParams {
  float color_albedo[3];    /* @0  */
  float color_specular[3];  /* @12 */
  float emission[3];        /* @24 */
  float smoothness;         /* @36 */
} urt_RayTraceParams;

typedef struct urt_MeshObject {
  float localToWorldMatrix[16]; /* @0  Unity Matrix4x4 memory order = column-major m[col*4+row] */
  int32_t indices_offset;       /* @64 first slot of this mesh in _Indices */
  int32_t indices_count;        /* @68 number of index slots (3 per triangle) */
  urt_RayTraceParams lighting;  /* @72 */
} urt_MeshObject;

typedef struct urt_Sphere {
  float position[3];            /* @0  */
  float radius;                 /* @12 */
  urt_RayTraceParams lighting;  /* @16 */
} urt_Sphere;

typedef struct urt_BVHNode {
  float vmin[3];                /* @0  */
  float vmax[3];                /* @12 */
  int32_t index;                /* @24 <0: interior/filler, >=0: object id (implicit heap 2i+1, 2i+2) */
} urt_BVHNode;
#pragma pack(pop)

#define URT_STRIDE_PARAMS 40
#define URT_STRIDE_MESHOBJECT 112
#define URT_STRIDE_SPHERE 56
#define URT_STRIDE_BVHNODE 28
#define URT_STRIDE_VEC3 12
#define URT_STRIDE_INDEX 4

#ifdef __cplusplus
}
static_assert(sizeof(urt_RayTraceParams) == URT_STRIDE_PARAMS, "RM:42");
static_assert(sizeof(urt_MeshObject) == URT_STRIDE_MESHOBJECT, "RM:43");
static_assert(sizeof(urt_Sphere) == URT_STRIDE_SPHERE, "RM:44");
static_assert(sizeof(urt_BVHNode) == URT_STRIDE_BVHNODE, "RM:45");
#endif
