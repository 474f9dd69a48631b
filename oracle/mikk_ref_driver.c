/* TEST INFRASTRUCTURE ONLY.  Flat-array driver around the reference's own third_party/MikkTSpace/mikktspace.c, compiled together with
 * it (from where it lies under /root/reference, see the Makefile) into oracle/_ref/libmikk_ref.so: the checker of
 * vk_gltf_renderer_amd/csrc/host/mikktspace_tangents.cpp (tests/test_mikktspace.py).  It feeds the library the way the reference
 * does (src/gltf_create_tangent.cpp:97-165: three vertices per face, position / normal / texcoord through the index list) and
 * records what m_setTSpaceBasic reports: tangent and sign per face corner. */
#include <mikktspace.h>

typedef struct
{
  const float *   pos, *nrm, *uv;
  const unsigned* idx;
  int             numTris;
  float*          out;
} Mesh;

static int  numFaces(const SMikkTSpaceContext* c) { return ((const Mesh*)c->m_pUserData)->numTris; }
static int  numVerts(const SMikkTSpaceContext* c, int f) { (void)c; (void)f; return 3; }
static void position(const SMikkTSpaceContext* c, float o[], int f, int v)
{
  const Mesh* m = (const Mesh*)c->m_pUserData;
  const float* p = m->pos + 3 * (unsigned long)m->idx[f * 3 + v];
  o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
}
static void normal(const SMikkTSpaceContext* c, float o[], int f, int v)
{
  const Mesh* m = (const Mesh*)c->m_pUserData;
  const float* p = m->nrm + 3 * (unsigned long)m->idx[f * 3 + v];
  o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
}
static void texcoord(const SMikkTSpaceContext* c, float o[], int f, int v)
{
  const Mesh* m = (const Mesh*)c->m_pUserData;
  const float* p = m->uv + 2 * (unsigned long)m->idx[f * 3 + v];
  o[0] = p[0]; o[1] = p[1];
}
static void basic(const SMikkTSpaceContext* c, const float t[], float sign, int f, int v)
{
  float* o = ((Mesh*)c->m_pUserData)->out + 4 * (unsigned long)(f * 3 + v);
  o[0] = t[0]; o[1] = t[1]; o[2] = t[2]; o[3] = sign;
}

__attribute__((visibility("default"))) int mikk_ref(const float* pos, const float* nrm, const float* uv, const unsigned* idx, int numTris, float* out)
{
  Mesh                 m = {pos, nrm, uv, idx, numTris, out};
  SMikkTSpaceInterface i = {0};
  SMikkTSpaceContext   c = {0};
  i.m_getNumFaces = numFaces; i.m_getNumVerticesOfFace = numVerts; i.m_getPosition = position; i.m_getNormal = normal;
  i.m_getTexCoord = texcoord; i.m_setTSpaceBasic = basic;
  c.m_pInterface = &i;
  c.m_pUserData  = &m;
  return genTangSpaceDefault(&c);
}
