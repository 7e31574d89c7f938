"""Host-side logic of the reference-shaped surface that needs no GPU: the train_run schedule pinned by
a fixture the REFERENCE's own train_run produced (tests/golden/make_golden2.py)."""
import numpy as np


def test_train_run_schedule_matches_reference(golden):
    """QAgent.train_run (r_learning.py:254-346): alpha decay every decay_step episodes (:293-294) and on every new top
    tile (:311-313), round(max(alpha * decay, low), 4) (:258), num_eps + 1 episodes (:284), ma100 history (:315-317),
    best-game capture (:299-306), the 1000-episode report (:318-341) — replayed with the outcomes of the reference's own
    1 010 episodes fed through `episode()`; the learning rate in force at every episode, the counters and the whole log
    must come out the same."""
    import game2048.r_learning as rl
    g = golden('train_schedule.npz')
    n, alpha, decay, decay_step, low = g['params']
    agent = rl.QAgent(name='sched', storage='local', console='local', n=int(n), alpha=float(alpha), decay=float(decay),
                      decay_step=int(decay_step), low_alpha_limit=float(low), with_weights=False)
    agent.top_tile = 6
    lines = []
    agent.print = lambda text='': lines.append(str(text))
    seen = dict(alpha=[], step=[], next_decay=[])
    outcomes = iter(zip(g['rows'], g['scores'], g['odometers']))

    def episode():                                            # what r_learning.py:224-252 hands back to train_run
        seen['alpha'].append(agent.alpha)
        seen['step'].append(agent.step)
        seen['next_decay'].append(agent.next_decay)
        row, score, odometer = next(outcomes)
        game = rl.Game(score=int(score), row=row.astype(np.int32))
        game.odometer = int(odometer)
        agent.step += 1
        return game
    agent.episode = episode
    agent.train_run(num_eps=len(g['scores']) - 1, saving=False)
    assert next(outcomes, None) is None                       # exactly num_eps + 1 episodes
    assert np.array_equal(np.array(seen['alpha']), g['alpha_at_start'])
    assert np.array_equal(np.array(seen['step']), g['step_at_start'])
    assert np.array_equal(np.array(seen['next_decay']), g['next_decay_at_start'])
    assert (agent.alpha, agent.step, agent.top_tile, agent.top_score, agent.next_decay) == \
        (float(g['final_alpha']), int(g['final_step']), int(g['final_top_tile']), int(g['final_top_score']), int(g['final_next_decay']))
    assert agent.top_game.score == int(g['top_game_score'])
    assert np.array_equal(np.array(agent.train_history), g['train_history'])
    keep = [ln for ln in lines if not ln.rstrip().endswith(' min') and not ln.startswith('Total time')]
    want = str(g['log']).split('\n')
    got = '\n'.join(keep).split('\n')
    assert got == want, next((i, a, b) for i, (a, b) in enumerate(zip(got + [None], want + [None])) if a != b)
