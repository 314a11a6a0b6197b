"""CPU oracle for the RobustART AddNoise hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``robustart_amd/`` may import this
package: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and only as the checker (never as the thing that is
measured or shipped).  The product path fails loudly when the HIP library is
missing; it never falls back to this code.

Parity status (see DESIGN.md "Oracle pinning"):
  * pinned against unmodified reference code run in the build container
    (fixtures under tests/golden/, generator tests/golden/make_golden.py):
    gaussian_noise, shot_noise, speckle_noise, contrast, fog, zoom_blur,
    pixelate, jpeg_compression, the AddNoise('imagenet-c') batch path,
    APGD-CE (Linf/L2), APGD-T, MIM.
  * parity UNPINNED (reference delegates to wheels that are absent here and has
    no tests of its own): impulse_noise, defocus_blur, glass_blur, motion_blur,
    snow, frost, brightness, elastic_transform, gaussian_blur, spatter, saturate
    (scikit-image / OpenCV / ImageMagick), pgd_linf / pgd_l2 / fgsm (foolbox
    3.3.1), pgd_l1 (ART).  These restate the upstream library semantics recalled
    in SURVEY.md Appendix B.
"""
