"""CPU oracle for the DeepMerge hot path.  TEST INFRASTRUCTURE ONLY.

This package is a plain PyTorch-CPU / numpy restatement of the reference's arithmetic for
the ShfitScaleFormer / ViT pair encoder, the contrastive loss, Adam, and the
ExtractFeatures pooling + region-adjacency similarity sweep.  Every function cites the
reference file:line it follows.

Rules (see DESIGN.md "Oracle"):
  * only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it;
  * it is the checker, never the thing measured or shipped -- nothing under `deepmerge_amd/`
    imports it, and the product path raises if the HIP library is missing;
  * it is pinned against golden vectors produced by the *unmodified reference modules*
    (tests/golden/make_golden.py, run in the build container where /root/reference exists;
    the vectors are committed under tests/golden/*.npz and checked by tests/test_oracle_*.py).

Parity status: PINNED for the model / loss / Adam arithmetic (reference imported and run on CPU) and, since
round 2, for the ExtractFeatures sweep and the loaders' window arithmetic: tests/golden/sweep.npz holds the outputs
of the reference's OWN `ExtractFeatures.Euclidean_distance` / `MC_Lyu_2020`, of its per-edge loop body
(np.concatenate + np.mean(axis=0) + distance + .max()) and of `MyUtils1.get_scales`,
`calculate_left_top_point_and_size`, `cut_image`, `get_designed_features` / `get_all_features` (h5py / osgeo / cv2
are absent; empty module objects satisfy the import statements, none of their functions is called).
tests/test_oracle_sweep.py and tests/test_oracle_patches.py check `oracle.sweep` (numpy + pinned-order C) and
`oracle.patches` against it; the GPU tests check the kernels against the same file.
Still NOT pinned: the OpenCV INTER_AREA band resize (cv2 is absent and unpinned upstream): `oracle.patches` states
"parity unpinned" for that one function; windows whose length equals the target (resize = identity) are pinned.
"""
