/*
 * equss_hip_dev.h - developer knobs of libequss_hip.so.  NOT part of the drop-in ABI.
 *
 * Each one switches which kernel instance the entry points of equss_hip.h select, for tests that
 * compare two instances of the same operation and for tuning experiments.  They set process-wide
 * variables without any synchronisation: call them only from the thread that issues the launches,
 * between launches, and restore the default afterwards.  The reference has no counterpart.
 */
#ifndef EQUSS_HIP_DEV_H
#define EQUSS_HIP_DEV_H

#ifdef __cplusplus
extern "C" {
#endif

/* 0 (default): screened fp16 search where it applies; 1: exhaustive fp32 search for every shape;
 * 2: as 0, but K = 256 / 512 on the twelve- / six-wave kernels that score every codeword twice instead of keeping the scores. */
void equss_dev_pq_mode(int mode);
/* equss_gemm_bf16 instance: 3 (default) stationary-weights / producer-consumer / large-tile ring kernel where each applies;
   6 the same without the stationary-weights kernel; 4 large-tile ring kernel only; 0..2 older 128x128 kernels. */
void equss_dev_gemm_variant(int variant);
/* large-tile GEMM ring: bits 0..2 = stages (2 default, 3, 4), bit 3 = prefer the 256x256 tile. */
void equss_dev_gemm_stages(int stages);
/* K step of the producer / consumer GEMM's ring: 64 (two 64-KiB slots, default) or 32 (four 32-KiB slots). */
void equss_dev_ws_bk(int bk);
/* CUs a large-tile GEMM launch may occupy (1..256, default 256). */
void equss_dev_gemm_cu_share(int cus);
/* waves per attention workgroup: 0 (default) = 4, 8 = the 8-wave variant. */
void equss_dev_attn_diag(int diag);
/* ViT-S blocks: 2 (default) proj + MLP in one kernel, 1 MLP in one kernel, 0 one launch per GEMM. */
void equss_dev_mlp_fused(int on);
/* largest number of trailing 128-row tiles that leave the fused kernel for the per-GEMM route. */
void equss_dev_mlp_tail_max(int tiles);
/* smallest number of 128-row tiles from which the fused kernels are used (default 384). */
void equss_dev_fuse_min_tiles(int tiles);
/* 1 (default): the fused block-tail kernel reads Wp / W1 as ring-stage images written at the start of the forward; 0: row-major. */
void equss_dev_tail_images(int on);
/* 1 (default): inside equss_vit_forward the producer / consumer block-tail kernel (64-row tiles, csrc/block_ws.hip);
 * 2: the same with four consumer waves of 64 rows instead of eight of 32;
 * 0: the eight-wave kernel of equss_block_tail_bf16 on the whole rounds of 128-row tiles + small-tile launches for the rest;
 * 3: the register-streamed kernel (csrc/block_rs.hip: weights global -> registers in fragment order, activations in LDS). */
void equss_dev_block_ws(int on);
/* block_rs kernel: 1 (default) first tiles of different heights per workgroup class, so that tile boundaries do not coincide. */
void equss_dev_rs_stagger(int on);

#ifdef __cplusplus
}
#endif
#endif /* EQUSS_HIP_DEV_H */
