"""The reference's open-domain evaluation loop over this package's ``MIPS``
(/root/reference/eval_phrase_retrieval.py: ``evaluate`` :49-91, ``evaluate_results`` :94-205; question loading
``open_utils.load_qa_pairs`` :104-160; metrics ``densephrases/utils/eval_utils.py`` :9-86): batches of
``eval_batch_size`` questions through ``mips.search``, top-k answers / evidences / titles / scores / char spans, EM and
F1 at 1 and at k, top-k redundancy, and the same per-question prediction record the reference dumps as ``*.pred``.

The query encoder stays the caller's (``query2vec``: list[str] -> list[(start [1,768], end [1,768], tokens)], or a
[B,1536] GPU tensor).  Pinned against the reference's own functions by tests/test_eval_loop.py
(oracle/make_golden_eval.py).
"""
from __future__ import annotations

import json
import re
import string
from collections import Counter
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np


# ----------------------------------------------------------------------------------------- eval_utils.py:9-86
def normalize_answer(s: str) -> str:
    s = "".join(ch for ch in s.lower() if ch not in set(string.punctuation))
    s = re.sub(r"\b(a|an|the)\b", " ", s)
    return " ".join(s.split())


def f1_score(prediction: str, ground_truth: str) -> float:
    p, g = normalize_answer(prediction), normalize_answer(ground_truth)
    if p in ("yes", "no", "noanswer") and p != g:
        return 0.0
    if g in ("yes", "no", "noanswer") and p != g:
        return 0.0
    pt, gt = p.split(), g.split()
    same = sum((Counter(pt) & Counter(gt)).values())
    if same == 0:
        return 0.0
    precision, recall = same / len(pt), same / len(gt)
    return 2 * precision * recall / (precision + recall)


def exact_match_score(prediction: str, ground_truth: str) -> bool:
    return normalize_answer(prediction) == normalize_answer(ground_truth)


def regex_match_score(prediction: str, pattern: str) -> bool:
    try:
        compiled = re.compile(pattern, flags=re.IGNORECASE + re.UNICODE + re.MULTILINE)
    except BaseException:
        return False
    return compiled.match(prediction) is not None


def max_over_ground_truths(metric_fn, prediction, ground_truths):
    return max(metric_fn(prediction, gt) for gt in ground_truths)


# ----------------------------------------------------------------------------------------- open_utils.py:104-160
def load_qa_pairs(data_path: str, do_lower_case: bool = False, q_idx: Optional[int] = None, draft: bool = False,
                  draft_num_examples: int = 100, truecase=None):
    q_ids, questions, answers, titles = [], [], [], []
    data = json.load(open(data_path))["data"]
    for data_idx, item in enumerate(data):
        if q_idx is not None and data_idx != q_idx:
            continue
        q_id = item["id"]
        if "origin" in item:
            q_id = item["origin"].split(".")[0] + "-" + q_id
        question = item["question"]
        if "[START_ENT]" in question:                       # KILT entity linking: a window around the mention
            question = question[max(question.index("[START_ENT]") - 300, 0):question.index("[END_ENT]") + 300]
        if len(item["answers"]) == 0:
            continue
        q_ids.append(q_id)
        questions.append(question)
        answers.append(item["answers"])
        titles.append(item.get("titles", [""]))
    questions = [q[:-1] if q.endswith("?") else q for q in questions]
    if do_lower_case:
        questions = [q.lower() for q in questions]
    if draft:
        q_ids, questions, answers, titles = (x[:draft_num_examples] for x in (q_ids, questions, answers, titles))
    if truecase is not None:
        questions = [truecase.get_true_case(q) if q == q.lower() else q for q in questions]
    return q_ids, questions, answers, titles


def _embed_all(questions: Sequence[str], query2vec: Callable, batch_size: int = 64):
    """eval_phrase_retrieval.py:33-46.  Returns a list of per-batch query arrays / tensors."""
    outs = []
    for i in range(0, len(questions), batch_size):
        o = query2vec(list(questions[i:i + batch_size]))
        if hasattr(o, "is_cuda"):
            outs.append(o)
        else:
            start = np.concatenate([np.asarray(x[0]) for x in o], 0)
            end = np.concatenate([np.asarray(x[1]) for x in o], 0)
            outs.append(np.concatenate([start, end], 1))
    if outs and not hasattr(outs[0], "is_cuda"):
        return np.concatenate(outs, 0)
    import torch
    return torch.cat(outs, 0)


# ----------------------------------------------------------------------------------------- eval_phrase_retrieval.py:49-205
def evaluate(mips, query2vec: Callable, test_path: str, top_k: int = 10, eval_batch_size: int = 64,
             max_answer_length: int = 10, aggregate: bool = False, agg_strat: str = "opt1", return_sent: bool = False,
             nprobe: int = 256, regex: bool = False, candidates: Optional[set] = None, do_lower_case: bool = False,
             truecase=None, pred_path: Optional[str] = None):
    """Returns (exact_match_top1, f1_score_top1, exact_match_topk, f1_score_topk, pred_out) -- the reference's four
    numbers (:205) and the dict it writes to ``<name>_<total>_top<k>.pred`` (:170-180, 197-203)."""
    qids, questions, answers, _ = load_qa_pairs(test_path, do_lower_case=do_lower_case, truecase=truecase)
    query_vec = _embed_all(questions, query2vec)
    predictions, evidences, titles, scores, se_poss = [], [], [], [], []
    on_device = hasattr(query_vec, "is_cuda")
    for i in range(0, len(questions), eval_batch_size):
        kw = dict(q_texts=questions[i:i + eval_batch_size], top_k=top_k, max_answer_length=max_answer_length,
                  aggregate=aggregate, agg_strat=agg_strat, return_sent=return_sent)
        result = (mips.search_device(query_vec[i:i + eval_batch_size], **kw) if on_device
                  else mips.search(query_vec[i:i + eval_batch_size], nprobe=nprobe, **kw))
        predictions += [[r["answer"] for r in out][:top_k] if len(out) > 0 else [""] for out in result]
        evidences += [[r["context"] for r in out][:top_k] if len(out) > 0 else [""] for out in result]
        titles += [[r["title"] for r in out][:top_k] if len(out) > 0 else [[""]] for out in result]
        scores += [[r["score"] for r in out][:top_k] if len(out) > 0 else [-1e10] for out in result]
        se_poss += [[(r["start_pos"], r["end_pos"]) for r in out][:top_k] if len(out) > 0 else [(0, 0)] for out in result]

    if candidates is not None:                                                   # :96-107
        topk_preds = [list(filter(lambda x: (x in candidates) or (x.lower() in candidates), a)) for a in predictions]
        predictions = [a[:top_k] if len(a) > 0 else [""] for a in topk_preds]
    else:
        predictions = [a[:top_k] if len(a) > 0 else [""] for a in predictions]
    top1_preds = [a[0] for a in predictions]

    em_topk_sum = em_top1_sum = f1_topk_sum = f1_top1_sum = 0.0
    pred_out: Dict[str, dict] = {}
    match_fn = regex_match_score if regex else exact_match_score
    for i in range(len(predictions)):
        em_topk = max(max_over_ground_truths(match_fn, p, answers[i]) for p in predictions[i][:top_k])
        em_top1 = max_over_ground_truths(match_fn, top1_preds[i], answers[i])
        rd_topk = sum(max_over_ground_truths(match_fn, p, [predictions[i][0]]) for p in predictions[i][:top_k])
        f1_topk = f1_top1 = 0
        if not regex:
            f1_topk = max(max_over_ground_truths(f1_score, p, answers[i]) for p in predictions[i][:top_k])
            f1_top1 = max_over_ground_truths(f1_score, top1_preds[i], answers[i])
        em_topk_sum += em_topk
        em_top1_sum += em_top1
        f1_topk_sum += f1_topk
        f1_top1_sum += f1_top1
        assert len(predictions[i]) <= top_k
        pred_out[qids[i]] = {
            "question": questions[i], "answer": answers[i], "prediction": predictions[i], "score": scores[i],
            "title": titles[i], "evidence": evidences[i], "em_top1": bool(em_top1), f"em_top{top_k}": bool(em_topk),
            "f1_top1": f1_top1, f"f1_top{top_k}": f1_topk, "se_pos": se_poss[i], "rd_topk": rd_topk,
        }
    total = max(len(predictions), 1)
    out = (100.0 * em_top1_sum / total, 100.0 * f1_top1_sum / total, 100.0 * em_topk_sum / total, 100.0 * f1_topk_sum / total)
    if pred_path is not None:
        with open(pred_path, "w") as f:
            json.dump(pred_out, f)
    return out + (pred_out,)
