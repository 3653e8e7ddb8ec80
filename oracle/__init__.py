"""CPU oracle for the GANSpace component-discovery hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``ganspace_amd/`` may import this
package: it is the *checker* for the HIP path (``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg are the
only legal importers).  The product path fails loudly when the HIP library is
missing; it never falls back to this code.

Parity status: **pinned by generated fixtures, unpinned by reference tests.**
The reference repository (harskish/ganspace) ships no test, golden vector or
known-answer file for ``decomposition.py`` / ``estimators.py`` (SURVEY.md §4,
§8c); the arithmetic lives in the third-party dependency scikit-learn
(``IncrementalPCA``; unpinned in the reference's ``environment.yml:15``, 1.7.2
in this image).  The oracle is therefore pinned two ways:

* ``tests/golden/*.npz`` hold outputs of the *reference's own*
  ``estimators.IPCAEstimator`` (imported from /root/reference with the absent
  ``fbpca`` stubbed) and of the in-tree ``models/stylegan/model.py:G_mapping``
  on seeded inputs; ``tests/golden/make_golden.py`` is the generating script.
* scikit-learn's own known-answer tests for ``IncrementalPCA`` are re-run
  against the restatement in ``tests/test_oracle.py``.

Modules
-------
ipca       restatement of sklearn ``IncrementalPCA.partial_fit`` (SVD form), of
           the d x d Gram recurrence the HIP path implements, and of exact
           covariance PCA.
zstream    the z-stream / block-index protocol of ``decomposition.compute``.
synth      numpy restatement of the StyleGAN2 mapping network and the BigGAN
           ``gen_z`` linear layer (random-init synthetic generators).
"""
