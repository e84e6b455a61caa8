"""Option surface of the reference's config_auto4rec.get_param (GURU/config_auto4rec.py:5-237).

Every attribute the hot path reads (`param.<attr>`) keeps its name and derivation.  What the
reference hard-codes and BASELINE.json's configurations need to change is exposed through extra,
optional attributes of `args` (seq_len, vocab_size_a/b, users_a/b, overlap_users, n_blocks, dropout,
data_path, d_ff_model, steps_tune); when they are absent the reference values apply.
"""
import os

_BLOCKS = {1: 3, 2: 3, 3: 6, 4: 6, 5: 6, 6: 6}                                   # :44-53
_DATASETS = {1: "book_movie", 2: "sport_cloth", 3: "business_10", 4: "business_30",
             5: "business_50", 6: "business_75"}                                 # :54-61
_DOMAINS = {1: ("movie", "book"), 2: ("sport", "cloth")}                         # :63-69 (3-6: wesee/txvideo)
# dataset -> (data_path, vocab_a, users_a, vocab_b, users_b, overlap users)      # :111-139
_PUBLIC = {
    "sport_cloth": ("/data/ceph/seqrec/data_guru/public/Amazon_torch/sport_cloth", 11835, 9024, 42139, 46810, 1062),
    "book_movie": ("/data/ceph/seqrec/data_guru/public/Amazon_torch/movie_book", 5536, 4261, 51366, 42940, 584),
}
_BUSINESS = {"10": (194756, 1203194, 1434999, 288155, 1311440), "30": (585580, 1398645, 1630372, 303942, 1356831),
             "50": (976556, 1594256, 1825737, 316338, 1397521), "75": (1464731, 1838276, 2069892, 329679, 1438803)}


def _opt(args, name, default):
    v = getattr(args, name, None)
    return default if v is None else v


class get_param(object):
    def __init__(self, args, make_dirs=True):
        self.date = args.date
        self.pad_index = 0
        self.num_train_neg = 5
        self.n_bpr_neg = 5
        self.enc_maxlen = int(_opt(args, "seq_len", 100))          # reference: 100 (:27-28)
        self.rec_maxlen = self.enc_maxlen
        self.d_ff = int(_opt(args, "d_ff_model", 512))             # reference: 512 (:31); --d_ff only feeds rs_d_ff
        self.d_model = args.d_model
        self.num_heads = args.n_head
        self.rs_hidden_units = 128
        self.d_k = 32
        self.d_v = 32
        self.dataset_pick = args.dataset_pick
        self.target_domain = args.target_domain
        self.num_blocks = int(_opt(args, "n_blocks", _BLOCKS[self.dataset_pick]))
        self.rs_d_model, self.rs_num_blocks, self.rs_d_ff = args.d_model, self.num_blocks, args.d_ff
        self.n_negs = args.n_negs
        self.dataset = _DATASETS[self.dataset_pick]
        names = _DOMAINS.get(self.dataset_pick, ("wesee", "txvideo"))
        self.domain_name_a, self.domain_name_b = names
        self.domain_name = names[0] if self.target_domain == "a" else names[1]
        self.decoder_neg = args.decoder_neg
        self.fixed_enc = args.fix_enc
        self.lr_rs = args.lr
        self.batch_size = args.batch_size
        self.batch_size_val = args.batch_size_val
        business = "business" in self.dataset
        cross = args.cross == "True"
        if cross or business:
            sub = "%s_%s_%d_%d_mg" % (self.dataset, self.domain_name, args.d_model, args.run)
        else:
            sub = "%s_%d_%d_mg" % (self.domain_name, args.d_model, args.run)
        self.result_path = os.path.join(args.result_path, sub)
        if business:
            rate = self.dataset.split("_")[1]
            self.data_path = "/data/ceph/seqrec/data/business/kdd_%s_f" % rate
            over, ua, ub, va, vb = _BUSINESS[rate]
        else:
            self.data_path, va, ua, vb, ub, over = _PUBLIC[self.dataset]
        self.data_path = _opt(args, "data_path", self.data_path)
        self.num_overlap_users = int(_opt(args, "overlap_users", over))
        self.num_users_a = int(_opt(args, "users_a", ua))
        self.num_users_b = int(_opt(args, "users_b", ub))
        self.vocab_size_a = int(_opt(args, "vocab_size_a", va)) + 1     # = EOS id of domain a (Q10)
        self.vocab_size_b = int(_opt(args, "vocab_size_b", vb)) + 1
        if self.target_domain == "a":
            self.vocab_size, self.num_users = self.vocab_size_a, self.num_users_a
        else:
            self.vocab_size, self.num_users = self.vocab_size_b, self.num_users_b
        self.logdir = os.path.join(self.result_path, self.dataset)
        self.model_path = os.path.join(self.result_path, "model")
        if make_dirs:
            os.makedirs(self.model_path, exist_ok=True)

        sas = args.sas == "True"
        if business:                                                     # :157-177
            self.batch_size_over = 100
            if sas:
                self.training_steps = self.training_steps_tune = 3000
            else:
                self.training_steps = 2000
                self.training_steps_tune = 3000 if cross else 2000
                self.n_warmup_steps = 1000
                if cross:
                    self.batch_size_over = max(10, int(self.batch_size * float(rate) / 100))
        else:                                                            # :178-201
            per = int(self.num_users / self.batch_size)
            if sas:
                self.training_steps = 200 * per + 1
                self.training_steps_tune = 600 * per + 1
                self.batch_size_over = 100
            elif not cross:
                self.training_steps = self.training_steps_tune = 500 * per + 1
                self.n_warmup_steps = 1000
            else:
                users_n = self.num_users_a + self.num_users_b
                self.training_steps = 300 * int(users_n / self.batch_size) + 1
                self.training_steps_tune = 400 * int(users_n / self.batch_size) + 1
                self.eval_step = int(users_n / (self.batch_size * 5))
                self.n_warmup_steps = int(self.training_steps / 2)
                ratio = int(users_n / max(1, self.num_overlap_users))
                self.batch_size_over = max(10, int(self.batch_size / ratio) + 1)
        self.freq_train_ep = {"movie": 400, "cloth": 200, "book": 200}.get(self.domain_name, 350)   # :203-212
        if business:
            self.eval_epoch = 70
            self.eval_steps = int(50000 / self.batch_size_val) + 1
            self.candidate_size = 19999
        else:
            self.eval_epoch = 10
            self.eval_steps = int(2000 / self.batch_size_val) + 1
            self.candidate_size = 199
        self.dropout_rate = float(_opt(args, "dropout", 0.5))           # reference: 0.5 (:225)
        self.smoothing = 0.1
        self.dis_dim = self.d_model * 5
        self.training_steps_tune = int(_opt(args, "steps_tune", 300))   # ":237  # for testing"
