"""CPU oracle for the CaDM CEM-planning / ensemble-training hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``cadm_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg use it, and only as the checker / reported baseline.

PARITY UNPINNED (strict sense): the reference (younggyoseo/CaDM, TensorFlow 1.15)
cannot run as a whole in this image (tensorflow, gym, mujoco_py, baselines are
absent) and ships no tests, golden vectors or fixtures for this path
(SURVEY.md §4, §8c).  The arithmetic itself lives in third-party TensorFlow
1.15.0 kernels (tf.matmul, tf.nn.softplus, tf.nn.top_k, tf.random.*,
AdamOptimizer) whose published semantics are restated here; every function
cites the reference call site (file:line under /root/reference) it follows.
What IS pinned to reference code executed in the build container: the planner
graph's data movement, member assignment, CEM update, variable creation order
and draw order (tests/golden/make_graph_golden.py runs cadm/dynamics/core/utils.py
unchanged on a numpy-eager tensorflow stand-in -> graph_golden.npz), the training
loss composition and the model's variable list (tests/golden/make_loss_golden.py
runs the reference model's constructor on the same stand-in -> loss_golden.npz) and the
future-window builder (tests/golden/make_f2_golden.py runs the reference's own
process_samples -> f2_windows.npz).

What else pins the oracle (tests/test_oracle_*.py):
  * a LITERAL transcription of the reference graph (tile / transpose / reshape
    chain, including index quirks Q1/Q2) checked against an independent
    INDEX-MAPPED formulation;
  * fp64 evaluation as mathematical truth next to the fp32 evaluation;
  * analytic known-answer cases (zero weights, sigma=0 stats, top-k ties, ...);
  * torch.autograd (fp64) for the training-step gradients;
  * the TF1 Adam closed form;
  * Random123 known-answer vectors for Philox4x32-10.
"""
