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

Parity status: PINNED for the model / loss / Adam arithmetic (reference imported and run
on CPU).  The ExtractFeatures sweep (`oracle.sweep`) is restated from source because the
reference module needs h5py/osgeo (absent); its distance helper is a pure-numpy function
whose text is restated here and pinned by known-answer cases in tests/test_oracle_sweep.py.
The OpenCV INTER_AREA resize used by the reference's data loaders is NOT pinned
(cv2 is absent and unpinned upstream): `oracle.patches` states "parity unpinned" itself.
"""
