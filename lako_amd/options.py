"""Flag system of the reader drivers — same flag names, types and defaults as the reference's
`src/options.py` (base :90-112, optim :20-48, reader :55-66, eval :50-53) so that the argument lists of
run_okvqa_train.sh / run_okvqa_test.sh parse unchanged; `add_retriever_options` carries the flags train_retriever.py reads
(src/options.py:68-86; the index-building ones are parsed and unused).  Added flags (all optional): --synthetic, --dtype, --steps."""
from __future__ import annotations

import argparse


class Options:
    def __init__(self):
        self.parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
        self._base()

    def _base(self):
        p = self.parser
        p.add_argument("--name", type=str, default="experiment_name")
        p.add_argument("--checkpoint_dir", type=str, default="./checkpoint/")
        p.add_argument("--model_path", type=str, default="none")
        p.add_argument("--per_gpu_batch_size", default=1, type=int)
        p.add_argument("--maxload", type=int, default=-1)
        p.add_argument("--local_rank", type=int, default=-1)
        p.add_argument("--main_port", type=int, default=-1)
        p.add_argument("--seed", type=int, default=0)
        p.add_argument("--eval_freq", type=int, default=500)
        p.add_argument("--save_freq", type=int, default=5000)
        p.add_argument("--eval_print_freq", type=int, default=1000)
        # additions of this implementation
        p.add_argument("--synthetic", type=str, default=None, metavar="B,N,L,T",
                       help="train on synthetic OKVQA-shaped batches of this shape (no tokenizer / data files needed)")
        p.add_argument("--dtype", type=str, default="bf16", choices=["bf16", "f32"], help="compute dtype of the kernels")
        p.add_argument("--steps", type=int, default=None, help="stop after this many optimizer steps")
        p.add_argument("--tokenizer", type=str, default=None, help="local T5 tokenizer directory (default t5-<model_size>)")

    def add_optim_options(self):
        p = self.parser
        p.add_argument("--gpu", default=0, type=int)
        p.add_argument("--epochs", default=1000, type=int)
        p.add_argument("--early_stop", default=30, type=int)
        p.add_argument("--dataset", default="okvqa", type=str)
        p.add_argument("--stream", default=1, type=int)
        p.add_argument("--use_fact", default="yes", type=str)
        p.add_argument("--fact_use_way", default="concate", type=str)
        p.add_argument("--attention_score_style", default="mean", type=str)
        p.add_argument("--consider_context_attention", default="no", type=str)
        p.add_argument("--use_last_half_layer_attention", default="no", type=str)
        p.add_argument("--ans_attention", default="no", type=str)
        p.add_argument("--version", default="v1", type=str)
        p.add_argument("--asymmetric_retri", default="no", type=str)
        p.add_argument("--warmup_steps", type=int, default=1000)
        p.add_argument("--total_steps", type=int, default=1000)
        p.add_argument("--scheduler_steps", type=int, default=None)
        p.add_argument("--accumulation_steps", type=int, default=1)
        p.add_argument("--dropout", type=float, default=0.1)
        p.add_argument("--lr", type=float, default=1e-4)
        p.add_argument("--clip", type=float, default=1.0)
        p.add_argument("--optim", type=str, default="adam")
        p.add_argument("--scheduler", type=str, default="fixed")
        p.add_argument("--weight_decay", type=float, default=0.1)
        p.add_argument("--fixed_lr", action="store_true")
        return self

    def add_eval_options(self):
        p = self.parser
        p.add_argument("--write_results", action="store_true")
        p.add_argument("--write_crossattention_scores", action="store_true")
        return self

    def add_reader_options(self):
        p = self.parser
        p.add_argument("--train_data", type=str, default="none")
        p.add_argument("--eval_data", type=str, default="none")
        p.add_argument("--model_size", type=str, default="base")
        p.add_argument("--use_checkpoint", action="store_true")
        p.add_argument("--text_maxlength", type=int, default=100)
        p.add_argument("--answer_maxlength", type=int, default=-1)
        p.add_argument("--no_title", action="store_true")
        p.add_argument("--n_context", type=int, default=1)
        # not a reference flag: the reference IS transformers 3.0.2 (README.md:21), where the decoder's first cross-attention layer owns a
        # relative-position table and every cross-attention adds its bias (src/model.py:301-303,323-329).  Set it to train / evaluate a
        # checkpoint of the reference with those semantics; without it the table is ignored, as every transformers >= 4 does
        p.add_argument("--legacy_cross_bias", action="store_true")
        return self

    def add_retriever_options(self):
        p = self.parser
        p.add_argument("--indexing_batch_size", type=int, default=50000)
        p.add_argument("--n-subquantizers", type=int, default=0)
        p.add_argument("--n-bits", type=int, default=8)
        p.add_argument("--passages_embeddings", type=str, default="fact_embedding_dim256_at_11-07-13.pkl")
        p.add_argument("--save_or_load_index", action="store_true")
        p.add_argument("--train_data", type=str, default="none")
        p.add_argument("--eval_data", type=str, default="none")
        p.add_argument("--indexing_dimension", type=int, default=256)
        p.add_argument("--no_projection", action="store_true")
        p.add_argument("--question_maxlength", type=int, default=130)
        p.add_argument("--passage_maxlength", type=int, default=130)
        p.add_argument("--no_question_mask", action="store_true")
        p.add_argument("--no_passage_mask", action="store_true")
        p.add_argument("--extract_cls", action="store_true")
        p.add_argument("--no_title", action="store_true")
        p.add_argument("--n_context", type=int, default=1)
        p.add_argument("--retriever_layers", type=int, default=12, help="(addition) BERT layers of a randomly initialised retriever")
        return self

    def parse(self, argv=None):
        return self.parser.parse_args(argv)
