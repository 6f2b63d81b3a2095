"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatement (plain PyTorch fp32 / numpy float64, no reference imports) of
the LeftRefill diffusion-sampling hot path: SD2-inpainting UNet forward +
DDIM/CFG sampler (+ the re-arranged multi-view self-attention, the multi-condition
consistency sampler and the training objective p_losses, whose autograd is the
checker of the HIP backward kernels).

Who may import this package: `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` -- as the checker / reported baseline, never
as the product.  `leftrefill_amd/` must NOT import it; the product path fails
loudly when the HIP extension is missing.

Parity pin: the restatement is checked against golden vectors produced by
importing the real reference (`/root/reference`, CPU, fp32, vanilla attention)
with `oracle/make_golden.py` (needs `oracle/ref_import.py` stubs; runs only in
the authoring container).  The vectors live in `tests/golden/*.npz`:
ops / unet / multiview / sampler / sampler_multi (operators, whole UNets, DDIM
trajectories), train (the reference's loss and d loss / d context through its own
CheckpointFunction), vae / vae_hip (the KL-VAE, generated from the reference's
modules directly -- there is no restatement of the VAE, the goldens are compared
with the drop-in modules and the HIP path).
The reference itself ships no tests / golden vectors for this path
(SURVEY.md section 4), so the pin is "restatement == reference outputs run
here", with torch 2.10 CPU kernels standing in for the reference's
torch 1.13.1 + xformers 0.0.16 CUDA kernels.
"""
