"""CPU oracle for the CreamFL contrastive hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in ``creamfl_amd/`` may import this
package: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and only as the checker / the timed CPU baseline.

Every function here is a plain torch-CPU (or numpy) restatement of one row of
SURVEY.md section 8(a); its docstring cites the reference file:line it follows
(paths relative to the FLAIR-THU/CreamFL checkout).  The reference is pure
Python on PyTorch, so the restatement is Python on PyTorch too (there is no
C/C++ source to compile into ``oracle/_ref``).

Pinning: ``tests/golden/make_golden.py`` imports the reference's own modules
from ``/root/reference`` (build container only), runs them on seeded inputs and
stores inputs + outputs as ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``
checks every function here against those vectors.  Rows whose reference code
is an inline loop body that cannot be imported in isolation (A3/A4/A5) are
pinned against the literal statement sequence evaluated with the reference's
imported criterion object (``src.losses.create('softmax')``).
"""
from .pair_loss import (pair_loss_literal, pair_loss_closed_form,
                        pair_loss_grads_closed_form, match_prob)
from .bank_contrast import (inter_contrast, intra_contrast, client_contrast_loss,
                            mm_client_contrast_loss,
                            client_contrast_grads_closed_form)
from .conw import conw_logprob, conw_weights, conw_aggregate
from .pie import (pie_attention_pool, pie_head, l2_normalize,
                  image_head_glue)
from .client_encoders import resnet_client_forward, text_client_forward, pcme_towers_forward
from .kd import kd_loss, code_sim
from .recall import recall_ranks_literal, recall_ranks_count, recall_scores

__all__ = [n for n in dir() if not n.startswith('_')]
