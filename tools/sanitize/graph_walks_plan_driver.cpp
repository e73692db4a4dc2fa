#include <cstdio>
#include <cstring>
#include "fie.h"
int main() {
    fie_unet_config u;
    memset(&u, 0, sizeof(u));
    u.batch = 2; u.latent_h = 128; u.latent_w = 128; u.text_len = 77; u.num_blocks = 3;
    u.block_out_channels[0] = 320; u.block_out_channels[1] = 640; u.block_out_channels[2] = 1280;
    u.layers_per_block = 2;
    u.down_attn[1][0] = u.down_attn[1][1] = 2; u.down_attn[2][0] = u.down_attn[2][1] = 4;
    u.up_attn[0][0] = u.up_attn[0][1] = u.up_attn[0][2] = 4; u.up_attn[1][0] = u.up_attn[1][1] = u.up_attn[1][2] = 2;
    u.mid_attn = 4; u.mid_resnets = 2; u.head_dim = 64; u.norm_num_groups = 32; u.norm_eps = 1e-5f;
    u.cross_attention_dim = 2048; u.addition_time_embed_dim = 256; u.pooled_dim = 1280;
    u.num_cond_channels = 4; u.cond_channels[0] = 16; u.cond_channels[1] = 32; u.cond_channels[2] = 96; u.cond_channels[3] = 256;
    printf("unet ws %lld, controlnet ws %lld, residuals %d\n", (long long)fie_unet_workspace_bytes(&u), (long long)fie_controlnet_workspace_bytes(&u), fie_unet_num_residuals(&u));
    for (int b = 1; b <= 16; b *= 2) { u.batch = b; printf("batch %d: %lld\n", b, (long long)fie_unet_workspace_bytes(&u)); }
    u.batch = 2; u.mid_resnets = 1; u.mid_attn = 0; printf("mid 1: %lld\n", (long long)fie_unet_workspace_bytes(&u));
    fie_vae_config v;
    memset(&v, 0, sizeof(v));
    v.latent_h = 128; v.latent_w = 128; v.num_blocks = 4; v.block_out_channels[0] = 128; v.block_out_channels[1] = 256; v.block_out_channels[2] = v.block_out_channels[3] = 512;
    v.layers_per_block = 2; v.norm_num_groups = 32; v.norm_eps = 1e-6f; v.out_channels = 3;
    printf("vae dec %lld enc %lld\n", (long long)fie_vae_decode_workspace_bytes(&v, 128, 128), (long long)fie_vae_encode_workspace_bytes(&v));
    for (int h = 1; h <= 64; h *= 4) printf("vae dec %d: %lld\n", h, (long long)fie_vae_decode_workspace_bytes(&v, h, 3 * h));
    fie_clip_config c = {2, 77, 1280, 20, 32, 5120, 1280, 0, 1e-5f};
    printf("clip %lld\n", (long long)fie_clip_text_workspace_bytes(&c));
    c.projection_dim = 0; c.layers = 1; printf("clip one layer no proj %lld\n", (long long)fie_clip_text_workspace_bytes(&c));
    return 0;
}
