"""Interactive front-end with the reference's HTTP contract (gui.py:1-47), served by Flask:

    GET  /                  -> the page (web_ui.html next to this file)
    POST /execute_function  {"variable1": "<meta prompt>"}  -> runs `run.execute` for one random seed on the model that
                            `run.setup` loaded, copies the PNG to static/output.png, answers {"result": "<image path>"}
    POST /post              -> echo of the submitted form (debug endpoint of the reference)

One generation at a time: the request thread calls `execute` against the module-global state of
`utils.shared_state`, exactly as the reference does — and unlike the reference this is enforced: the captured hipGraphs
replay on static device buffers, so a second request that arrives while one is running is answered 409 "busy" instead of
running concurrently (Flask's development server handles requests on threads).  `run.main` starts this when `--interactive true` is given.
Off the hot path (nothing here is timed or accelerated); the page is a small stand-in written for this build —
boxes are typed as fractions, the meta-prompt grammar is the reference's (`[phrase:x,y,w,h]`, `[phrase:x,y]`).
"""
import random
import shutil
import threading
from pathlib import Path

from flask import Flask, jsonify, render_template, request

from .utils import shared_state

HERE = Path(__file__).resolve().parent
app = Flask(__name__, template_folder=str(HERE), static_folder=str(HERE / "static"))


_busy = threading.Lock()   # held for the duration of one generation


def _execute(config):
    from . import run
    return run.execute(config)


@app.after_request  # cache-breaker: the output image keeps its URL
def add_no_store_header(response):
    response.headers["Cache-Control"] = "no-store"
    return response


@app.route("/", methods=["GET"])
def index():
    return render_template("web_ui.html")


@app.route("/execute_function", methods=["POST"])
def execute_function():
    meta_prompt = request.json["variable1"]
    if not _busy.acquire(blocking=False):
        return jsonify({"error": "busy: a generation is running"}), 409
    try:
        shared_state.config.meta_prompt = meta_prompt
        shared_state.config.seeds = [int(random.randrange(4294967294))]
        print(meta_prompt)
        image_path = _execute(shared_state.config)
        (HERE / "static").mkdir(exist_ok=True)
        shutil.copyfile(str(image_path), str(HERE / "static" / "output.png"))
    finally:
        _busy.release()
    return jsonify({"result": str(image_path)})


@app.route("/post", methods=["POST"])
def post():
    return "recived: {}".format(request.form)


def run(host="127.0.0.1", port=5000):
    app.run(host=host, port=port, debug=False)


if __name__ == "__main__":
    run()
