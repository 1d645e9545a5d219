"""CPU oracle for the GraspLDM grasp-generation hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``graspldm_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` do, and there only as the checker / reported baseline.

Contents
--------
point_ops.c / cpu_backend.py  scalar C restatement of the reference's nine
                              forward CUDA kernels behind the 12 ``_backend``
                              names (self-pinned: the reference has no CPU path)
schedulers.py                 DDIM / DDPM step restated from the public
                              algorithm (diffusers is absent here: PARITY
                              UNPINNED at that third-party boundary)
torch_ref.py                  functional fp32 torch-CPU restatement of the
                              encoder / denoiser / decoder / sampler graph,
                              pinned by tests/golden/*.npz captured from the
                              reference's own Python (oracle/make_golden.py)
ref_import.py, make_golden.py container-only: import /root/reference with
                              shims and write the golden fixtures
"""
