#include "mi_host.h"
#include <cstdio>
int main(int argc, char** argv)
{
  for(int i = 1; i < argc; ++i)
  {
    MiScene* s = nullptr;
    if(mi_scene_load(argv[i], &s) == MI_PT_OK && s)
    {
      (void)mi_scene_num_triangles(s);
      const int n = mi_scene_num_animations(s);
      for(int a = 0; a < n && a < 2; ++a)
        mi_scene_update_animation(s, a, 0.5f);
      mi_scene_cut_alpha(s, 4);
      mi_scene_recompute_tangents(s, 1, 1);
      MiCamera cam;
      mi_scene_camera(s, 0, &cam);
      mi_scene_destroy(s);
    }
    printf("ok %s\n", argv[i]);
    fflush(stdout);
  }
  return 0;
}
